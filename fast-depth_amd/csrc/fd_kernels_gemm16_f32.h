// fd_kernels_gemm16_f32.h -- second-generation fp32 pointwise GEMM for gfx950: v_mfma_f32_16x16x4_f32, one workgroup per CU,
// the K tile split between the two waves of a SIMD.
//
//   out[M][N] = act(A[M][K] * Wt[N][K32]^T + bias[N])          (pointwise 1x1 conv + folded BN + ReLU/ReLU6;
//                                                                reference imagenet/mobilenet.py:35-37, models.py:70-75)
//
// Why a second kernel (measurements: DESIGN.md section 3): the 32x32x2 kernel of fd_kernels_f32.h quantises the work into 32x32
// tiles per SIMD -- at batch 32 every pointwise layer has 3.06 of them per SIMD, so a quarter of the machine idles in the last
// round -- and every one of its (3 per CU) workgroups pays prologue and epilogue at the same moment.  Here
//   * the MFMA is 16x16x4 (32-cycle issue, 40-cycle dependent latency, 4 accumulator registers): the quantum is a 16x16 tile;
//   * a workgroup owns  m_stride x 64  outputs, m_stride <= 16*TM chosen by the host so that the grid is (a multiple of) the
//     256 CUs: M = 6272 -> m_stride 196 (TM = 13, 49 of 52 tile slots useful = 94 %) instead of 3.06 -> 4 rounds (76 %);
//   * 8 waves: wave (kh, wn) owns the 16-column block wn of all TM row tiles for the k-half kh of every 32-deep K tile.  The two
//     waves that share a SIMD are the two k-halves of the same columns: perfectly balanced, they interleave on the matrix pipe,
//     and their partial sums meet once, in the epilogue, through LDS;
//   * per K tile a wave issues TM+1 ds_read_b128 (A fragments are shared by the 4 column waves, 16 bytes = 4 consecutive k per
//     lane feed 4 MFMA steps: the K-permutation trick of fd_pw_gemm_f32) and 4*TM MFMAs on TM independent accumulators; the
//     fragments of tile t+1 are requested WHILE the MFMAs of tile t are issued (double-buffered in registers);
//   * operands go L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into an S-stage ring issued S-1 tiles ahead; since a whole K tile's
//     fragments sit in registers, a stage is free again as soon as every wave has read it: one s_barrier per K tile;
//   * leader / follower: an LDS-DMA instruction blocks the issuing wave for ~100 cycles.  Measured (tools/microbench/gemm16.hip,
//     ablations): when both waves of a SIMD issue their share of the DMA, the slower one pays its share uncovered after its partner
//     has reached the barrier (+0.24 us per K tile on 1.32).  So the kh = 0 waves ("leaders", s_setprio 1) issue ALL the DMA pieces,
//     spread through their MFMA stream, and win the matrix-pipe arbitration; the followers fill every slot the leaders leave and
//     finish the K tile with a pure MFMA + ds_read tail;
//   * epilogue through LDS: partial sums are exchanged in the output tile's own [row][col] image, which is then written with
//     16-byte stores of whole 256-byte rows.
// LDS rows are 128 bytes with the XOR swizzle chunk' = chunk ^ ((row >> 1) & 7), applied to the DMA source side and to the
// fragment reads (conflict-free for the 16-lane groups of ds_read_b128: rows r..r+15 x chunks c..c+3).
//   * fused consumer: where a workgroup's m_stride rows are WHOLE frames of the stored output (14x14 = 196 rows, 7x7 = 49) the depthwise
//     layer that consumes this output -- per-channel, so it needs exactly this workgroup's 64 columns of exactly these frames -- is
//     evaluated in the epilogue, from the LDS image of the tile, and ITS output is what leaves the kernel: no launch, no HBM round trip
//     for the intermediate tensor (3x3 stride 1 / 2, 5x5, and 5x5 on the nearest-x2 upsampling of the tile: FDW template parameter).
#pragma once
#include "fd_device.h"
#include "fd_bn_stats.h"

// the depthwise layer fused behind a pointwise GEMM (fd_pw_gemm16_f32<..., FDW = its kernel size>)
struct fd_dwfuse {
    const float *w;            // folded taps, tap-major [K*K][C]  (C = the GEMM's N)
    const float *b;            // folded bias [C]
    float *out;                // its output, NHWC [frames][Ho][Wo][C]
    int H, W;                  // frame size of the GEMM's stored output (m_stride % (H*W) == 0)
    int S, up;                 // stride (1 | 2); up = 1: the input is the nearest-x2 upsampling of the frame (S == 1)
    float hi;                  // activation: clamp(0, hi), hi = 6 (ReLU6) or +inf (ReLU)
    int store_pw;              // also store the GEMM's own output (plans that keep every layer's activations)
};

#ifdef FD_GEMM16_PROBE
__device__ long long fd_gemm16_probe[4 * 4096];          // measurement aid (tools/microbench/gemm16.hip): shader-clock and 100 MHz timestamps per workgroup
#endif

// Train-mode forward (TRAIN = the PRODUCER's activation, FD_ACT_RELU_ / FD_ACT_RELU6_; 0 = inference): the same kernel with
//   * A = the producer's RAW output: every A fragment becomes act(a * s[k] + t[k]) right before its first MFMA (the producer's BatchNorm table,
//     zero beyond K, sits in LDS behind the ring; a lane's four k of a K tile are the same for all TM fragments: two ds_read_b128 per K tile);
//   * Wt = the LIVE weights [N][K] (K % 32 == 0: pitch K32 = K), no bias, no activation: the RAW conv output leaves the kernel;
//   * the epilogue adds the per-column partial statistics of the tile's valid rows to the unit's statistics rows (fd_stat_add);
//   * tr.fin.rows != null: the PRODUCER's BatchNorm is finalised here -- the four follower waves (kh = 1: idle while the leaders issue the prologue's
//     LDS-DMA) derive the [2][K] table from the producer's statistics rows; workgroup 0 is the writer.
// Replaces fd_pw_gemm_train_f32 (the 32x32x2 structure) on the units one round of workgroups covers -- the 14x14 / 7x7 maps at batch 32.
#ifndef FD_G16_TRAIN_AHEAD
#define FD_G16_TRAIN_AHEAD 2       // (0 / 2 / 4 measured equal: 42.4 / 42.2 / 42.5 us on the 512 x 512 units -- the cost of the transform is its instruction count, not its latency)
#endif
struct fd_g16_train {
    const float *st;           // the producer's table [4][K] (scale, shift, mean, invstd) when it was finalised by a launch of its own
    fd_stat_rows sr;           // statistics rows of THIS unit
    fd_bn_fin fin;             // rows != null: the producer's finalisation runs in this kernel
};

// ABL (measurement aid, 0 in the product): 1 = no LDS-DMA in the steady state (stages keep the first tiles), 2 = also no fragment reads,
// 3 = also no per-tile barrier, 4 = full K loop but no global stores -- wrong results, used by tools/microbench/gemm16.hip to price each
// ingredient of the kernel.
template <int TM, int STAGES, int ACT, int ABL = 0, int FDW = 0, int TRAIN = 0>
__global__ void __launch_bounds__(512)
fd_pw_gemm16_f32(const float *__restrict__ A, const float *__restrict__ Wt, const float *__restrict__ bias,
                 float *__restrict__ out, int M, int N, int K, int K32, int m_stride, int m_tiles, int n_tiles, const fd_dwfuse fz, const fd_g16_train tr)
{
    static_assert(TRAIN == 0 || (FDW == 0 && ACT == 0), "train mode: raw output, no fused consumer");
    constexpr int BM = TM * 16, BN = 64, BK = 32, ROWS = BM + BN;
    constexpr int STAGE = ROWS * BK;                        // floats per stage
    constexpr int NG = ROWS / 8;                            // LDS-DMA row groups (8 rows = 1 KiB) per stage
    constexpr int RG = (NG + 3) / 4;                        // per leader wave (a leader without an own group in the last round repeats its first)
    constexpr int OP = BN + 4;                              // row pitch (floats) of the output tile image in LDS
    static_assert(STAGES * STAGE >= BM * OP + (TRAIN != 0 ? 8 * 2 * 64 : 0), "the output tile image (+ the statistics scratch) must fit the ring");
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 3, kh = wave >> 2;                 // waves w and w+4 share a SIMD (waves are dealt to the SIMDs cyclically)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % n_tiles, mt = (slot / n_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * m_stride;
    const int n0 = nt * BN;
    const bool leader = kh == 0;
#ifdef FD_GEMM16_PROBE
    long long pc0 = 0, pw0 = 0;
    if (tid == 0) { pc0 = clock64(); pw0 = wall_clock64(); }
#endif

    // ---- LDS-DMA sources (leaders only).  Everything that changes from K tile to K tile is wave-uniform (base pointer + t * 128 bytes,
    // LDS destination), everything per-lane is loop-invariant (a 32-bit byte offset of the lane's row / swizzled chunk), so that a
    // piece costs two or three scalar instructions and the DMA itself (global_load_lds_dwordx4 voffset, saddr): the first version
    // re-derived a 64-bit per-lane address with ~10 vector instructions per piece, and those issue slots were the K loop's overhead.
    const int wn_u = FD_UNIFORM(wn);
    unsigned src_off[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        int g = wn_u + 4 * i;
        if (g >= NG) g = wn_u;                               // duplicate of this wave's first group: identical bytes to the identical place
        const int r = g * 8 + (lane >> 3);
        const int chunk4 = ((lane & 7) ^ ((r >> 1) & 7)) * 4;
        if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src_off[i] = (unsigned)((row * K + chunk4) * 4); }
        else { int row = n0 + (r - BM); if (row > N - 1) row = N - 1; src_off[i] = (unsigned)(((long)row * K32 + chunk4) * 4); }
    }
    const bool ragged_k = K != K32;
    auto piece = [&](int t, int i) {
        int g = wn_u + 4 * i;
        if (g >= NG) g = wn_u;
        const bool is_a = g < BM / 8;                         // a row group is all A rows or all weight rows (BM % 8 == 0)
        const char *base = reinterpret_cast<const char *>(is_a ? A : Wt) + (size_t)t * (BK * 4);
        const char *p = base + src_off[i];
        if (ragged_k && is_a && t == K32 / BK - 1) {         // ragged K, last tile: chunks beyond K read this row's first chunk instead (any
            const int chunk4 = ((lane & 7) ^ (((g * 8 + (lane >> 3)) >> 1) & 7)) * 4;          // finite data: the padded weights are 0 there)
            if (t * BK + chunk4 >= K) p -= (t * BK + chunk4) * 4;
        }
        fd_glds16(reinterpret_cast<const float *>(p), smem + (t % STAGES) * STAGE + g * 8 * BK);
    };

    // ---- accumulators: the leader starts at the folded-BN bias of its column, the follower at 0 ----
    const int col = n0 + wn * 16 + (lane & 15);
    const float bv = (TRAIN == 0 && leader && col < N) ? bias[col] : 0.0f;
    fd_f32x4 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { acc[i].x = bv; acc[i].y = bv; acc[i].z = bv; acc[i].w = bv; }

    // ---- fragment offsets (floats) within a stage: row tile i sits i * 16 rows further down with the SAME swizzle ((16 i) >> 1 is a
    // multiple of 8), so one address register + immediate offsets serve all TM reads ----
    const int chunk = kh * 4 + (lane >> 4);
    const int a_off0 = (lane & 15) * BK + ((chunk ^ (((lane & 15) >> 1) & 7)) << 2);
    const int rowb = BM + wn * 16 + (lane & 15);
    const int b_off = rowb * BK + ((chunk ^ ((rowb >> 1) & 7)) << 2);

    const int T = K32 / BK;
    fd_f32x4 fa[2][TM], fb[2];
    float *tab = smem + STAGES * STAGE;                      // TRAIN: [2][K32] scale / shift of the producer (0 beyond K)
    fd_f32x4 sv[2], tv[2];                                   // ... of this lane's four k of the K tile in fragment buffer 0 / 1
    // (requested now, stored to LDS after the prologue's LDS-DMA has been issued: the table's round trip runs under the first tiles' -- the loads are OLDER
    // than every DMA piece, so the prologue's counted vmcnt wait covers them)
    constexpr int TABQ = 2;                                  // K32 <= 1024 (checked by the plan)
    float tsc[TABQ], tsh[TABQ];
    if (TRAIN != 0 && !tr.fin.rows) {
#pragma unroll
        for (int j = 0; j < TABQ; ++j) {
            const int k = tid + 512 * j, kc = k < K ? k : 0;
            const float a = tr.st[kc], b = tr.st[K + kc];     // FD_ST_SCALE = 0, FD_ST_SHIFT = 1 (fd_kernels_train.h)
            tsc[j] = k < K ? a : 0.0f; tsh[j] = k < K ? b : 0.0f;
        }
    }
    // One K tile: 4*TM MFMAs on the fragments in buffer B with the loads spread through the MFMA stream: the leader's DMA pieces of
    // tile t+STAGES first, then (both roles) the fragment reads of tile t+1.  The uniform branches also pin the instruction order.
    auto step = [&](auto buf, auto role, int t) {
        constexpr int B = decltype(buf)::value;
        constexpr bool LEAD = decltype(role)::value != 0;
        constexpr int NDMA = LEAD ? RG : 0, NL = NDMA + TM + 1, SP = (4 * TM) / NL;
        static_assert(SP >= 1, "not enough MFMA slots for the loads");
        const bool do_frags = ABL != 2 && ABL != 3 && t + 1 < T, do_dma = ABL != 1 && ABL != 2 && ABL != 3 && t + STAGES < T;
        if (ABL != 3 && t + 1 < T) {
            if (LEAD) { if ((ABL == 0 || ABL == 4) && t + STAGES - 1 < T) fd_wait_vmcnt<(STAGES - 2) * RG>(); else fd_wait_vmcnt<0>(); }   // tile t+1 has landed
            fd_block_barrier_lds();                          // ... for every leader, and every wave has finished reading tile t's fragments
        }
        const float *nxa = smem + ((t + 1) % STAGES) * STAGE + a_off0, *nxb = smem + ((t + 1) % STAGES) * STAGE + b_off;
        if (TRAIN != 0 && do_frags) { sv[1 - B] = fd_ld4(tab + (t + 1) * BK + chunk * 4); tv[1 - B] = fd_ld4(tab + K32 + (t + 1) * BK + chunk * 4); }
#pragma unroll
        for (int idx = 0; idx < 4 * TM; ++idx) {
            const int q = idx / TM, i = idx % TM;
            // BatchNorm + activation of the producer on the A fragments, FD_G16_TRAIN_AHEAD MFMAs before a fragment's first use (its VALU latency never meets the matrix pipe)
            if (TRAIN != 0 && idx == 0) {
#pragma unroll
                for (int j = 0; j < FD_G16_TRAIN_AHEAD && j < TM; ++j) fa[B][j] = fd_act4<TRAIN>(fa[B][j] * sv[B] + tv[B]);
            }
            if (TRAIN != 0 && q == 0 && i + FD_G16_TRAIN_AHEAD < TM) fa[B][i + FD_G16_TRAIN_AHEAD] = fd_act4<TRAIN>(fa[B][i + FD_G16_TRAIN_AHEAD] * sv[B] + tv[B]);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[B][i][q], fb[B][q], acc[i], 0, 0, 0);
            if ((idx + 1) % SP == 0) {
                const int l = (idx + 1) / SP - 1;            // load slot
                if (l < NDMA) {
                    if (do_dma) piece(t + STAGES, l);        // stage t % STAGES is free again: tile t+STAGES goes there
                } else if (l < NDMA + TM) {
                    if (do_frags) fa[1 - B][l - NDMA] = fd_ld4(nxa + (l - NDMA) * 16 * BK);
                } else if (l == NDMA + TM) {
                    if (do_frags) fb[1 - B] = fd_ld4(nxb);
                }
            }
        }
    };

    if (leader) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s)
            if (s < T) {
#pragma unroll
                for (int i = 0; i < RG; ++i) piece(s, i);
            }
        if (T >= STAGES) fd_wait_vmcnt<(STAGES - 1) * RG>(); else fd_wait_vmcnt<0>();
    }
    if (TRAIN != 0) {
        if (tr.fin.rows) {
            if (!leader) fd_stat_table_all<256>(tr.fin, K, K32, tid - 256, blockIdx.x == 0, [&](int k, float a, float b) { tab[k] = a; tab[K32 + k] = b; });
        } else {
#pragma unroll
            for (int j = 0; j < TABQ; ++j) { const int k = tid + 512 * j; if (k < K32) { tab[k] = tsc[j]; tab[K32 + k] = tsh[j]; } }
        }
        fd_block_barrier_lds();
    } else {
        fd_block_barrier();
    }
    {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = fd_ld4(smem + a_off0 + i * 16 * BK);
        fb[0] = fd_ld4(smem + b_off);
        if (TRAIN != 0) { sv[0] = fd_ld4(tab + chunk * 4); tv[0] = fd_ld4(tab + K32 + chunk * 4); }
    }
    if (leader) {
#ifndef FD_EMU
        __builtin_amdgcn_s_setprio(1);
#endif
        for (int t = 0; t < T; t += 2) {
            step(fd_int<0>{}, fd_int<1>{}, t);
            if (t + 1 < T) step(fd_int<1>{}, fd_int<1>{}, t + 1);
        }
#ifndef FD_EMU
        __builtin_amdgcn_s_setprio(0);
#endif
    } else {
        for (int t = 0; t < T; t += 2) {
            step(fd_int<0>{}, fd_int<0>{}, t);
            if (t + 1 < T) step(fd_int<1>{}, fd_int<0>{}, t + 1);
        }
    }

    // ---- epilogue: the two k-halves meet in the [row][col] image of the output tile in LDS.  Every wave deposits the partial sums of
    // the row tiles its partner finishes (leader: [0, H), follower: [H, TM)); the owner adds its own, applies the activation in
    // place, and the finished image leaves as whole 256-byte rows, 16 bytes per lane.
    // With a fused depthwise consumer (FDW) the image is laid out as zero-bordered frames -- pixel (y, x) of frame f sits in image row
    // f*PH*PW + (y + P)*PW + (x + P), P = the consumer's padding in stored pixels -- so that its taps are branch-free reads at constant
    // offsets; `rowmap` translates a tile row (= pixel number) into its image row. ----
    constexpr int H = (TM + 1) / 2;
    constexpr int KD = FDW == 0 ? 1 : FDW;
    __shared__ int rowmap[FDW != 0 ? BM : 1];
    int rows = M - m0 < m_stride ? (int)(M - m0) : m_stride;
    if (rows > BM) rows = BM;
    const int fH = fz.H, fW = fz.W, fHW = fH * fW, us = (FDW != 0 && fz.up) ? 1 : 0;
    const int P = FDW == 0 ? 0 : (us ? (KD / 2 + 1) / 2 : KD / 2);           // border in stored pixels (the upsampled 5x5 reaches 1 stored pixel out)
    const int PW = fW + 2 * P, PHW = (fH + 2 * P) * PW;
    const int frames = FDW != 0 ? rows / fHW : 0;
    fd_block_barrier_lds();                                  // every wave is done with the ring (all tiles landed and read)
    if (FDW != 0) {
        const int img_rows = frames * PHW + 1;                // + a dump row for the tile's slack rows
        for (int i = tid; i < img_rows * (OP / 4); i += 512) fd_st4(smem + i * 4, fd_zero4());
        for (int t = tid; t < BM; t += 512) {
            const int fr = t / fHW, rem = t - fr * fHW, y = rem / fW, x = rem - y * fW;
            rowmap[t] = fr < frames ? fr * PHW + (y + P) * PW + (x + P) : frames * PHW;
        }
        __syncthreads();
    }
    int prow[TM][4];                                          // image row of D register r of row tile i: tile row i*16 + 4*(l>>4) + r
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = i * 16 + 4 * (lane >> 4) + r;
            prow[i][r] = FDW != 0 ? rowmap[t] : t;
        }
    float *img = smem + wn * 16 + (lane & 15);                // column l & 15 of this wave's 16-column block
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const bool mine = leader ? i < H : i >= H;
        if (!mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r) img[prow[i][r] * OP] = acc[i][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const bool mine = leader ? i < H : i >= H;
        if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { float *o = img + prow[i][r] * OP; o[0] = fd_act<ACT>(acc[i][r] + o[0]); }
        }
    }
    __syncthreads();
#ifdef FD_GEMM16_PROBE
    if (tid == 0) { fd_gemm16_probe[4 * blockIdx.x] = pc0; fd_gemm16_probe[4 * blockIdx.x + 1] = clock64(); fd_gemm16_probe[4 * blockIdx.x + 2] = pw0; fd_gemm16_probe[4 * blockIdx.x + 3] = wall_clock64(); }
#endif
    if (ABL == 4) return;
    const int c4 = (lane & 15) * 4;
    if ((FDW == 0 || fz.store_pw) && n0 + c4 < N) {          // N % 4 == 0: a lane's 4 columns are all inside or all outside
        for (int r = wave * 4 + (lane >> 4); r < rows; r += 32)
            fd_st4(out + (m0 + r) * N + n0 + c4, fd_ld4(smem + (FDW != 0 ? rowmap[r] : r) * OP + c4));
    }
    if (TRAIN != 0) {
        // partial statistics of the tile: column c = tid & 63, the rows dealt to 8 row lanes (fixed order), which meet behind the image in LDS
        float *red = smem + BM * OP;                         // [8][2][64]
        const int c = tid & 63, g = tid >> 6;
        float ssum = 0.0f, ssq = 0.0f;
        for (int r = g; r < rows; r += 8) { const float v = smem[r * OP + c]; ssum += v; ssq = fmaf(v, v, ssq); }
        red[(g * 2 + 0) * 64 + c] = ssum; red[(g * 2 + 1) * 64 + c] = ssq;
        __syncthreads();
        if (tid < 64 && n0 + tid < N) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) { a += red[(gg * 2 + 0) * 64 + tid]; b += red[(gg * 2 + 1) * 64 + tid]; }
            fd_stat_add<FD_STAT_FWD>(tr.sr, mt, N, 0, n0 + tid, a);
            fd_stat_add<FD_STAT_FWD>(tr.sr, mt, N, 1, n0 + tid, b);
        }
    }
    if (FDW != 0) {
        // ---- the consuming depthwise layer on the zero-bordered frames: work-item = 4 channels (cg) x every 32nd output pixel of a frame ----
        constexpr int PD = KD / 2;
        const int cg = tid & 15, pl = tid >> 4;
        const int c = n0 + cg * 4;
        if (c >= N) return;
        fd_f32x4 wv[KD * KD];
#pragma unroll
        for (int t = 0; t < KD * KD; ++t) wv[t] = fd_ld4(fz.w + (long)t * N + c);
        const fd_f32x4 b4 = fd_ld4(fz.b + c);
        const int S = fz.S;
        const int Ho = (fH << us) / S, Wo = (fW << us) / S, HWo = Ho * Wo;
        const long f0 = m0 / fHW;
        const int dy = 32 / Wo, dx = 32 - dy * Wo;
        for (int fr = 0; fr < frames; ++fr) {
            const float *img_f = smem + (long)fr * PHW * OP + cg * 4;
            int oy = pl / Wo, ox = pl - oy * Wo;                 // pixel pl, pl + 32, ...: walked without further divisions
            for (int op = pl; op < HWo; op += 32) {
                fd_f32x4 a4 = b4;
                if (us) {
                    // tap (ky, kx) of output (oy, ox) reads stored pixel ((oy - PD + ky) >> 1, (ox - PD + kx) >> 1): arithmetic shifts, border P = 1
                    int ry[KD], rx[KD];
#pragma unroll
                    for (int k = 0; k < KD; ++k) { ry[k] = (((oy - PD + k) >> 1) + P) * PW; rx[k] = ((ox - PD + k) >> 1) + P; }
#pragma unroll
                    for (int ky = 0; ky < KD; ++ky)
#pragma unroll
                        for (int kx = 0; kx < KD; ++kx) a4 += fd_ld4(img_f + (ry[ky] + rx[kx]) * OP) * wv[ky * KD + kx];
                } else {
                    const float *p0 = img_f + ((oy * S) * PW + ox * S) * OP;       // tap (0, 0): stored pixel (oy*S - PD, ox*S - PD) = image (oy*S, ox*S)
#pragma unroll
                    for (int ky = 0; ky < KD; ++ky)
#pragma unroll
                        for (int kx = 0; kx < KD; ++kx) a4 += fd_ld4(p0 + (ky * PW + kx) * OP) * wv[ky * KD + kx];
                }
                fd_f32x4 r4;
                r4.x = fminf(fmaxf(a4.x, 0.0f), fz.hi); r4.y = fminf(fmaxf(a4.y, 0.0f), fz.hi);
                r4.z = fminf(fmaxf(a4.z, 0.0f), fz.hi); r4.w = fminf(fmaxf(a4.w, 0.0f), fz.hi);
                fd_st4(fz.out + ((f0 + fr) * HWo + op) * N + c, r4);
                ox += dx; oy += dy;
                if (ox >= Wo) { ox -= Wo; ++oy; }
            }
        }
    }
}
