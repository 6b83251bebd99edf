// fd_kernels_gemm16_h16.h -- the one-workgroup-per-CU pointwise GEMM of fd_kernels_gemm16_f32.h for 16-bit storage (T = fp16 / bf16 activations
// and packed weights, fp32 accumulation): v_mfma_f32_16x16x32_{f16,bf16}, whole frames per workgroup, and the depthwise layer that consumes the
// output evaluated in the epilogue.
//
//   out[M][N] = act(A[M][K] * Wt[N][K64]^T + bias[N])          (pointwise 1x1 conv + folded BN + ReLU/ReLU6;
//                                                                reference imagenet/mobilenet.py:35-37, models.py:70-75)
//   fused consumer (FDW = 3 | 5): the next unit's depthwise conv + folded BN + activation (mobilenet.py:31-33, models.py:61-68) on the tile
//
// Why: with 16-bit operands the 14x14 / 7x7 pointwise layers are 7-12 us launches of which ~1.5 us is matrix work, each followed by an ~10 us
// depthwise launch that re-reads what was just written (profiles/r02: 18 GEMM + 13 depthwise launches, 0.26 of the step's roofline).  Here
//   * a workgroup (8 waves) owns m_stride x 64 outputs where m_stride is a whole number of stored frames (196 = 14x14, 98 / 196 = 2 / 4 x 7x7), so the
//     per-channel depthwise consumer needs exactly this workgroup's tile: it runs from the tile's zero-bordered image in LDS and only ITS output
//     leaves the kernel -- one launch and no HBM round trip instead of two launches and a write + read of the intermediate;
//   * a 128-byte LDS row holds 64 elements (BK = 64) and a lane's 16-byte chunk IS one 16x16x32 operand (8 consecutive k): wave (kh, wn) owns the
//     16-column block wn of all TM row tiles for the k-half kh of every K tile -- TM MFMAs and TM + 1 ds_read_b128 per tile and wave; the
//     two k-halves meet once, in the epilogue, through LDS (as in the fp32 kernel);
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4) into an S-stage ring, issued by the four kh = 0 waves, S - 1 tiles ahead; same
//     XOR swizzle chunk' = chunk ^ ((row >> 1) & 7) on the DMA source side and on the fragment reads; one s_barrier per K tile;
//   * ragged shapes (pruned plans: K, N multiples of 8 only): K tails read finite data against the zero-padded weight rows, N tails are
//     clamped on load and masked on store; m_stride need not be a multiple of 16.
// Epilogue and fused consumer are those of fd_pw_gemm16_f32 (fp32 image of the tile in LDS), with T-typed stores.
#pragma once
#include "fd_kernels_gemm16_f32.h"
#include "fd_kernels_h16.h"

#ifndef FD_EMU
__device__ __forceinline__ fd_f32x4 fd_mfma_16x16x32(fd_half, fd_u16x8 a, fd_u16x8 b, fd_f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fd_f16x8, a), __builtin_bit_cast(fd_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ fd_f32x4 fd_mfma_16x16x32(fd_bf16, fd_u16x8 a, fd_u16x8 b, fd_f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(fd_bf16x8_hw, a), __builtin_bit_cast(fd_bf16x8_hw, b), c, 0, 0, 0);
}
#else
inline fd_f32x4 fd_mfma_16x16x32(fd_half, fd_u16x8 a, fd_u16x8 b, fd_f32x4 c) { return hipemu_mfma_f32_16x16x32_f16(a, b, c); }
inline fd_f32x4 fd_mfma_16x16x32(fd_bf16, fd_u16x8 a, fd_u16x8 b, fd_f32x4 c) { return hipemu_mfma_f32_16x16x32_bf16(a, b, c); }
#endif

// the value a T-typed store of v would hold
__device__ __forceinline__ float fd_ld1_round(fd_half, float v) { return (float)(_Float16)v; }
__device__ __forceinline__ float fd_ld1_round(fd_bf16, float v) { return fd_bf16_to_f32(fd_f32_to_bf16(v)); }

template <typename T, int TM, int STAGES, int ACT, int FDW, int ALLW = 1>   // ALLW: every wave issues its share of the LDS-DMA pieces (the MFMA work per K tile is small: there is no matrix-pipe stream to protect)
__global__ void __launch_bounds__(512)
fd_pw_gemm16_h16(const T *__restrict__ A, const T *__restrict__ Wt, const float *__restrict__ bias, T *__restrict__ out,
                 int M, int N, int K, int K64, int m_stride, int m_tiles, int n_tiles, const fd_dwfuse fz, const int abl = 0)
{
    constexpr int BM = TM * 16, BN = 64, ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;                       // bytes per stage: 128-byte rows of 64 elements
    constexpr int NG = ROWS / 8;                            // LDS-DMA row groups (8 rows = 1 KiB) per stage
    constexpr int NLW = ALLW ? 8 : 4;                       // waves that issue LDS-DMA
    constexpr int RG = (NG + NLW - 1) / NLW;                // pieces per issuing wave and K tile
    constexpr int OP = BN + 4;                              // row pitch (floats) of the output tile image in LDS
    static_assert(STAGES * STAGE >= BM * OP * 4, "the output tile image must fit the ring");
    FD_DYN_SMEM(smem_raw);
    unsigned char *ring = smem_raw;
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 3, kh = wave >> 2;                 // waves w and w+4 share a SIMD: the two k-halves of the same 16 columns
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % n_tiles, mt = (slot / n_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * m_stride;
    const int n0 = nt * BN;
    const bool leader = kh == 0;                             // (bias and the row tiles [0, H) of the epilogue)
    const bool loader = ALLW ? true : leader;

    // ---- LDS-DMA sources (leaders): row group g = wn + 4 i holds rows 8g .. 8g+7; lane l brings the 16-byte chunk that lands in slot (row, l & 7) ----
    const int wn_u = FD_UNIFORM(ALLW ? wave : wn);            // first row group of this wave
    unsigned src_off[RG];                                    // byte offset of the lane's row / swizzled chunk within A or Wt
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        int g = wn_u + NLW * i;
        if (g >= NG) g = wn_u;                               // duplicate of this wave's first group: identical bytes to the identical place
        const int r = g * 8 + (lane >> 3);
        const int chunk8 = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src_off[i] = (unsigned)((row * K + chunk8) * 2); }
        else { int row = n0 + (r - BM); if (row > N - 1) row = N - 1; src_off[i] = (unsigned)(((long)row * K64 + chunk8) * 2); }
    }
    const bool ragged_k = K != K64;
    const int T_ = (abl & 1) ? 1 : K64 / 64;            // (abl: measurement aid, 0 in the product)
    auto piece = [&](int t, int i) {
        int g = wn_u + NLW * i;
        if (g >= NG) g = wn_u;
        const bool is_a = g < BM / 8;                         // a row group is all A rows or all weight rows (BM % 8 == 0)
        const char *p = reinterpret_cast<const char *>(is_a ? (const void *)A : (const void *)Wt) + (size_t)t * 128 + src_off[i];
        if (ragged_k && is_a && t == T_ - 1) {               // ragged K, last tile: chunks beyond K read this row's first chunk instead (finite data;
            const int chunk8 = ((lane & 7) ^ (((g * 8 + (lane >> 3)) >> 1) & 7)) * 8;          // the zero-padded weight rows annihilate it)
            if (t * 64 + chunk8 >= K) p -= (t * 64 + chunk8) * 2;
        }
        fd_glds16(reinterpret_cast<const float *>(p), reinterpret_cast<float *>(ring + (t % STAGES) * STAGE + g * 8 * 128));
    };

    // ---- the fused consumer's folded taps and bias (tap-major [K*K][64] + [64] floats) go to LDS now: their global-load latency hides under the
    // K loop instead of opening the epilogue ----
    constexpr int KD_ = FDW == 0 ? 1 : FDW;
    __shared__ __attribute__((aligned(16))) float s_dww[FDW != 0 ? (KD_ * KD_ + 1) * 64 : 4];
    if (FDW != 0) {
        for (int i = tid; i < (KD_ * KD_ + 1) * 16; i += 512) {
            const int t = i >> 4, c = n0 + (i & 15) * 4;
            fd_f32x4 v = fd_zero4();
            if (c < N) v = t < KD_ * KD_ ? fd_ld4(fz.w + (long)t * N + c) : fd_ld4(fz.b + c);
            fd_st4(s_dww + t * 64 + (i & 15) * 4, v);
        }
    }

    // ---- accumulators: the leader starts at the folded-BN bias of its column, the follower at 0 ----
    const int col = n0 + wn * 16 + (lane & 15);
    const float bv = (leader && col < N) ? bias[col] : 0.0f;
    fd_f32x4 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { acc[i].x = bv; acc[i].y = bv; acc[i].z = bv; acc[i].w = bv; }

    // ---- fragment offsets (bytes) within a stage: row tile i sits 16 i rows further down with the SAME swizzle ----
    const int chunk = kh * 4 + (lane >> 4);                  // this lane's 8 k of the wave's k-half
    const int a_off0 = (lane & 15) * 128 + ((chunk ^ (((lane & 15) >> 1) & 7)) << 4);
    const int rowb = BM + wn * 16 + (lane & 15);
    const int b_off = rowb * 128 + ((chunk ^ ((rowb >> 1) & 7)) << 4);

    if (loader) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < T_) {
#pragma unroll
                for (int i = 0; i < RG; ++i) piece(s, i);
            }
    }
    for (int t = 0; t < T_; ++t) {
        if (loader) {
            // tiles up to min(t + STAGES - 2, T_ - 1) are in flight; tile t has landed when at most the younger ones remain outstanding
            const int younger = (t + STAGES - 2 < T_ - 1 ? t + STAGES - 2 : T_ - 1) - t;
            if (younger >= 2) fd_wait_vmcnt<2 * RG>(); else if (younger == 1) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();
        }
        fd_block_barrier_lds();                              // tile t is visible to every wave; every wave has finished reading tile t - 1
        if (loader && t + STAGES - 1 < T_) {
#pragma unroll
            for (int i = 0; i < RG; ++i) piece(t + STAGES - 1, i);   // into the stage tile t - 1 occupied
        }
        const unsigned char *cur = ring + (t % STAGES) * STAGE;
        const fd_u16x8 fb = *reinterpret_cast<const fd_u16x8 *>(cur + b_off);
        fd_u16x8 fa[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const fd_u16x8 *>(cur + a_off0 + i * 16 * 128);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i] = fd_mfma_16x16x32(T{}, fa[i], fb, acc[i]);
    }

    // ---- epilogue: as fd_pw_gemm16_f32 (the two k-halves meet in the [row][col] fp32 image of the tile; with a fused depthwise consumer the image
    // is laid out as zero-bordered frames) ----
    constexpr int H = (TM + 1) / 2;
    constexpr int KD = FDW == 0 ? 1 : FDW;
    __shared__ int rowmap[FDW != 0 ? BM : 1];
    int rows = M - m0 < m_stride ? (int)(M - m0) : m_stride;
    if (rows > BM) rows = BM;
    const int fH = fz.H, fW = fz.W, fHW = fH * fW, us = (FDW != 0 && fz.up) ? 1 : 0;
    const int P = FDW == 0 ? 0 : (us ? (KD / 2 + 1) / 2 : KD / 2);
    const int PW = fW + 2 * P, PHW = (fH + 2 * P) * PW;
    const int frames = FDW != 0 ? rows / fHW : 0;
    fd_block_barrier_lds();                                  // every wave is done with the ring (all tiles landed and read)
    if (abl & 4) return;
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
            for (int r = 0; r < 4; ++r) {
                float *o = img + prow[i][r] * OP;
                // the value the STORED tensor holds (rounded to T): the fused consumer must see what an unfused consumer would read back
                o[0] = fd_ld1_round(T{}, fd_act<ACT>(acc[i][r] + o[0]));
            }
        }
    }
    __syncthreads();
    const int c4 = (lane & 15) * 4;
    if ((FDW == 0 || fz.store_pw) && n0 + c4 < N) {          // N % 4 == 0: a lane's 4 columns are all inside or all outside
        for (int r = wave * 4 + (lane >> 4); r < rows; r += 32)
            fd_st4(out + (m0 + r) * N + n0 + c4, fd_ld4(smem + (FDW != 0 ? rowmap[r] : r) * OP + c4));
    }
    if (FDW != 0 && !(abl & 2)) {
        // ---- the consuming depthwise layer on the zero-bordered frames: work-item = 4 channels (cg) x every 32nd output pixel of a frame ----
        constexpr int PD = KD / 2;
        const int cg = tid & 15, pl = tid >> 4;
        const int c = n0 + cg * 4;
        if (c >= N) return;
        fd_f32x4 wv[KD * KD];
#pragma unroll
        for (int t = 0; t < KD * KD; ++t) wv[t] = fd_ld4(s_dww + t * 64 + cg * 4);
        const fd_f32x4 b4 = fd_ld4(s_dww + KD * KD * 64 + cg * 4);
        const int S = fz.S;
        const int Ho = (fH << us) / S, Wo = (fW << us) / S, HWo = Ho * Wo;
        const long f0 = m0 / fHW;
        const int dy = 32 / Wo, dx = 32 - dy * Wo;
        T *dw_out = reinterpret_cast<T *>(fz.out);
        for (int fr = 0; fr < frames; ++fr) {
            const float *img_f = smem + (long)fr * PHW * OP + cg * 4;
            int oy = pl / Wo, ox = pl - oy * Wo;                 // pixel pl, pl + 32, ...: walked without further divisions
            for (int op = pl; op < HWo; op += 32) {
                fd_f32x4 a4 = b4;
                if (us) {
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
                fd_st4(dw_out + ((f0 + fr) * HWo + op) * N + c, r4);
                ox += dx; oy += dy;
                if (ox >= Wo) { ox -= Wo; ++oy; }
            }
        }
    }
}
