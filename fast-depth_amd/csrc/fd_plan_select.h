// fd_plan_select.h -- the inference plan's records (Layer, fd_plan), the bank-conflict replay that picks LDS patch pitches, the lifetime arena and the pointwise tile selection (the shape thresholds in here were measured at batch 32 / 64: DESIGN.md sections 3, 10, 11)
// (translation unit fd_api.hip; fd_train_fwd.hip includes it for choose_pw16 / FD_G16_STAGES; split out of fd_api.hip in round 4 -- the plan code was a 1 160-line monolith)
#pragma once
#ifndef FD_G16_STAGES
#define FD_G16_STAGES 4      // depth of fd_pw_gemm16_f32's LDS-DMA ring, issued STAGES - 1 K tiles ahead (build switch for tools/build_variant.py).  Round 4: 4 instead of
                             // 3 stages (139 KB at TM = 13: still one workgroup per CU, as designed) -- conv7.3 38.0 -> 36.9 us, conv13.3 43.4 -> 41.4, the B = 32 step -1.7 %
#endif
namespace {

// Row pitch (floats) of the [pixels][pitch] LDS patch images that the depthwise kernels read with ds_read_b128 from (strip of `strip` pixels,
// channel group) work-items.  A wave64 ds_read_b128 is served in four fixed 16-lane groups, one LDS cycle each when the group's 16-byte
// pieces cover the 64 banks once (MI355X_MICROARCH.md, LDS); with the round-1 pitch cb + 4 = 36 the four strips of a group sat 144 dwords
// apart = 16 banks, two of them on the same banks: measured 32-45 % of all LDS cycles were conflict cycles in the 5x5 kernels, whose LDS
// pipe is 83 % busy.  This replays the lane -> address map of the kernels' strip reads and returns the smallest conflict-free pitch.
int pick_patch_pitch(int cb, int tw, int tw_in, int stride, int strip = 4)
{
    static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                   {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int lanes_c = cb / 4, tws = std::max(1, tw / strip);
    int cbq = 0; while ((1 << cbq) < lanes_c) ++cbq;
    int best = cb + 4; long best_cycles = -1;
    for (int pitch = cb + 4; pitch <= cb + 36; pitch += 4) {
        long cycles = 0;
        for (int wave = 0; wave < 4; ++wave)
            for (int g = 0; g < 4; ++g) {
                int first_addr[64][4], n_addr[64];
                for (int b = 0; b < 64; ++b) n_addr[b] = 0;
                for (int k = 0; k < 16; ++k) {
                    const int tid = wave * 64 + grp[g][k], c4 = tid & (lanes_c - 1), pt = tid >> cbq;
                    const int oy = pt / tws, ox = (pt - oy * tws) * strip;
                    const int addr = ((oy * stride) * tw_in + ox * stride) * pitch + c4 * 4;
                    for (int dw = 0; dw < 4; ++dw) {
                        const int b = (addr + dw) & 63;
                        bool seen = false;
                        for (int q = 0; q < n_addr[b] && q < 4; ++q) seen |= first_addr[b][q] == addr;
                        if (!seen) { if (n_addr[b] < 4) first_addr[b][n_addr[b]] = addr; ++n_addr[b]; }
                    }
                }
                int worst = 1;
                for (int b = 0; b < 64; ++b) worst = std::max(worst, n_addr[b]);
                cycles += worst;
            }
        if (best_cycles < 0 || cycles < best_cycles) { best_cycles = cycles; best = pitch; }
        if (cycles == 16) break;                               // 4 waves x 4 groups x 1 cycle: conflict free
    }
    return best;
}

struct PwCfg { int wgm, wgn, tm, tn; };

struct Layer {
    fd_layer_desc d;
    int in_h = 0, in_w = 0;      // logical input size (after upsampling)
    int out_h = 0, out_w = 0;
    size_t out_off = 0, out_bytes = 0;   // activation arena
    size_t w_off = 0, w_bytes = 0, b_off = 0;   // packed weights / bias
    size_t w_elems = 0;          // unpadded weight element count (algorithmic bytes)
    bool to_output = false;      // writes the network output buffer directly
    bool head = false;           // Cout == 1 pointwise: fd_head_pw1
    bool pw_packed_t = false;    // packed weights are 16-bit (pointwise layers of a 16-bit plan)
    // dw tiling
    int cbq = 0, th = 0, tw = 0, tiles_x = 0, tiles_y = 0, mode = 0, pstr = 0;   // pstr: LDS patch row pitch in LDS elements (pick_patch_pitch)
    int dw_n = 4;                // channels per work-item of the LDS-tiled depthwise kernel (8: 16-bit plans, storage-typed patches)
    int dw5_cl = 0, dw5_cbs = 0, dw5_groups = 0, dw5_bh = 0;   // > 0: the row-walking pixel-pair kernel fd_dw5_rows (16-bit plans, 5x5 on up2 + skip): channel lanes per strip, channels per block, strip groups per row, rows per band
    size_t wpk_off = 0;          // ... its tap pairs (fd_pack_dw5_pairs), behind the folded fp32 taps
    int csplit = 0;              // concatenating consumer: channels [0, csplit) come from src, the rest from skip
    bool skipped = false;        // depthwise layer executed inside the following pointwise layer's fused kernel
    int fuse_next_dw = -1;       // pointwise layer (fd_pw_gemm16_f32): index of the depthwise consumer evaluated in its epilogue
    int fused_into = -1;         // depthwise layer: index of the pointwise layer whose kernel produces this layer's output
    int fused_dw = -1;           // pointwise layer: index of the depthwise layer that runs inside its fd_dwpw_f32 unit
    int fuse_head = -1;          // fd_dwpw_f32 unit: index of the 32 -> 1 pointwise head evaluated on its accumulators (that layer's fused_into = this one)
    bool dwpw = false;           // pointwise layer: fused_dw runs inside fd_dwpw_f32 (large maps: tile of pixels x all output channels)
    int dp_th = 0, dp_tw = 0 /* log2 of the tile width */, dp_tiles_x = 0, dp_wm = 0, dp_nt = 0, dp_nld = 0, dp_xcd = 0;
    bool dw_rows = false;        // register-window 3x3 kernel (fd_dw3_rows_f32) instead of the LDS-tiled one
    bool dw_rows8 = false;       // ... its 16-bit variant with eight channels per work-item (fd_dw3_rows8)
    // stem
    int chunk = 0;
    // pw
    PwCfg pw{};
    int m_tiles = 0, n_tiles = 0, w_pitch = 0;
    int pw16_tm = 0, pw16_stride = 0;   // > 0: fd_pw_gemm16_f32 (16x16x4 MFMA, one workgroup per CU) with TM row tiles and this M stride per workgroup
    size_t lds = 0;
    dim3 grid;
    std::string info, sym;
    double alg_bytes = 0, alg_flops = 0;
};

}  // namespace

struct fd_plan {
    std::vector<Layer> layers;
    int B = 0, H = 0, W = 0, dtype = 0;
    uint32_t flags = 0, tune = 0;    // public plan flags (include/fastdepth_hip.h) / private tuning mask (fd_tuning.h)
    size_t ws_bytes = 0, weights_bytes = 0;
    unsigned char *ws = nullptr;
    bool packed = false;
    double alg_bytes = 0, alg_flops = 0;
};

namespace {

// ---- lifetime-based arena ------------------------------------------------------------------------
struct FreeList {
    std::vector<std::pair<size_t, size_t>> blocks;   // (offset, size), sorted by offset
    size_t top = 0;
    size_t alloc(size_t bytes)
    {
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].second >= bytes) {
                size_t off = blocks[i].first;
                blocks[i].first += bytes;
                blocks[i].second -= bytes;
                if (blocks[i].second == 0) blocks.erase(blocks.begin() + i);
                return off;
            }
        // grow: extend a trailing free block if it touches the top
        if (!blocks.empty() && blocks.back().first + blocks.back().second == top) {
            size_t off = blocks.back().first;
            top = off + bytes;
            blocks.pop_back();
            return off;
        }
        size_t off = top;
        top += bytes;
        return off;
    }
    void release(size_t off, size_t bytes)
    {
        auto it = std::lower_bound(blocks.begin(), blocks.end(), std::make_pair(off, (size_t)0));
        it = blocks.insert(it, {off, bytes});
        if (it + 1 != blocks.end() && it->first + it->second == (it + 1)->first) { it->second += (it + 1)->second; blocks.erase(it + 1); }
        if (it != blocks.begin() && (it - 1)->first + (it - 1)->second == it->first) { (it - 1)->second += it->second; blocks.erase(it); }
    }
};

// ---- kernel selection ----------------------------------------------------------------------------
// Pointwise tile: the fp32 MFMA GEMM is compute-bound for most layers, so the tile is chosen to (a)
// not waste MFMA work on a ragged N, (b) give the 256 CUs at least ~2 workgroups each, (c) otherwise be
// as large as possible (fewer LDS/L2 bytes per flop).
PwCfg choose_pw(long M, int N)
{
    const PwCfg c128x128{2, 2, 2, 2}, c128x64{2, 2, 2, 1}, c64x128{2, 2, 1, 2}, c64x64{2, 2, 1, 1}, c128x32{4, 1, 1, 1};
    if (N <= 32) return c128x32;
    auto blocks = [&](const PwCfg &c) { return (long)ceil_div(M, c.wgm * c.tm * 32) * ceil_div(N, c.wgn * c.tn * 32); };
    auto waste = [&](const PwCfg &c) { int bn = c.wgn * c.tn * 32; return (double)(ceil_div(N, bn) * bn) / N; };
    // Measured on MI355X (tools/microbench/gemm_tiles.hip, round 1): with the fp32 MFMA at 64 cycles per instruction the 64x64 tile
    // (one 32x32 accumulator per wave, 4+ workgroups per CU) beats the larger tiles on every shape of this
    // network -- latency hiding across workgroups matters more than operand reuse.
    // Exception (same measurements): the 14x14 layers (M = 6272 at batch 32, N, K >= 256) run 13 % faster on 128x64 --
    // both shapes are bound by the same wave quantisation (3.06 32x32 tiles per SIMD), the larger tile halves the
    // L2 -> LDS bytes per flop.
    // Re-measured with the final kernel (tools/microbench/gemm_tiles.hip sweep over all 18 shapes of the network, batch 32): 6272x512x{256,512}
    // run fastest on 64x128 (37.9 us vs 39.6 on 128x64 vs 42.5 on 64x64), 25088x128x256 on 128x64 (23.2 vs 25.7); everything
    // else on 64x64.
    (void)blocks; (void)waste; (void)c128x128;
    if (M > 4096 && M <= 16384 && N >= 512) return c64x128;
    if (M > 16384 && M <= 32768 && N > 32 && N <= 128) return c128x64;
    return c64x64;
}

// Second-generation kernel (fd_kernels_gemm16_f32.h): worth it when ONE round of workgroups (one per CU) covers the layer with few
// idle tile slots -- then its 16x16 quantum removes the 3.06 -> 4 rounding of the 32x32 kernel.  Measured at batch 32
// (tools/microbench/gemm16.hip, profiles/r02): 6272x512x512 34.2 vs 38.1 us, 1568x1024x1024 40.1 vs 45.1, 6272x256x512 22.0 vs 23.2,
// 25088x128x256 23.0 vs 24.2; layers that need two rounds (25088x256x256) or have K < 256 gain nothing and keep the first kernel.
struct Pw16Cfg { int tm = 0, stride = 0; double score = 0; };
Pw16Cfg choose_pw16(long M, int N, int K, bool force)
{
    Pw16Cfg best;
    if (N % 4 || M <= 0) return best;
    if (force) {                                             // test mode: the largest row-tile count the layer can fill, balanced strides
        best.tm = M > 112 ? 13 : (M > 64 ? 7 : 4);
        const long mt = (M + 16 * best.tm - 1) / (16 * best.tm);
        best.stride = (int)((M + mt - 1) / mt); best.score = 1.0;
        return best;
    }
    const int nt = ceil_div(N, 64);
    for (int tm : {13, 7, 4}) {
        const long mtiles = std::max<long>(1, 256 / nt);      // the most M tiles one round can hold
        long stride = (M + mtiles - 1) / mtiles;
        if (stride > 16 * tm) continue;                      // would need a second round
        const long wgs = ((M + stride - 1) / stride) * nt;
        const double score = std::min(1.0, wgs / 256.0) * ((double)stride / (16 * tm));      // fraction of the chip's MFMA slots doing useful work
        if (score > best.score) { best.tm = tm; best.stride = (int)stride; best.score = score; }
    }
    if (best.score < 0.72 || K < 256) best = Pw16Cfg();
    return best;
}

int pw_lds_bytes(const PwCfg &c) { return FD_F32_STAGES * (c.wgm * c.tm * 32 + c.wgn * c.tn * 32) * 32 * 4; }   // 3-stage ring of 128-byte rows

int ilog2(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }
}  // namespace
