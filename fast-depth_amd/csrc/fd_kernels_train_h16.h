// fd_kernels_train_h16.h -- pointwise (1x1) kernels of the 16-bit TRAIN step: z, G and the GEMM operands are stored as
// T = fd_bf16 (fd_half also compiles; the C ABI only offers bf16 because fp16 gradients would need loss scaling), the
// matrix instructions are v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation, statistics / tables / parameter
// gradients / master weights stay fp32.
//
//   forward   z = act1(z_in*s+t) x W^T     fd_pw_gemm_train_h16   BN+activation of the producer applied to the A fragment
//                                                                 after the ds_read (fp32 math, re-packed to T)
//   backward  dz = BN-backward(G, z)       fd_bn_bwd_apply_h16    once per unit, IN PLACE over G: both backward GEMMs read
//                                                                 the same operand, so forming it on the fly would do the
//                                                                 fp32 table math twice per element per tile
//             G_in = mask * (dz x W)       fd_pw_dgrad_h16        B operand = W^T stored [K][N64]
//             dW   = dz^T x a_in           fd_pw_wgrad_h16        the reduction index (pixels) is the ROW index of both
//                                                                 operands in memory: tiles are transposed on their way
//                                                                 into LDS (16-byte global reads, 2-byte LDS writes), so
//                                                                 that fragments are single ds_read_b128
// The per-step conversion of the fp32 master weights into the two 16-bit operand layouts is fd_pack_train_w_h16.
#pragma once
#include "fd_kernels_h16.h"
#include "fd_kernels_bwd.h"

// (fd_unpack8 / fd_pack8 / fd_ld8 / fd_st8: fd_kernels_h16.h)

// ------------------------------------------------------------------------------------------------
// Master weights W[N][K] (fp32, torch layout) -> wt[N][K64] and wtt[K][N64] in T, zero padded along the reduction index of
// the GEMM that reads them (forward: k, backward-data: n).
// ------------------------------------------------------------------------------------------------
// One launch converts the weights of up to FD_PACK_MAX pointwise units (blockIdx.y = unit; the records travel as a kernel argument).
#define FD_PACK_MAX 24
template <typename T> struct fd_pack_rec { const float *w; T *wt; T *wtt; int N, K, K64, N64; };
template <typename T> struct fd_pack_table { fd_pack_rec<T> rec[FD_PACK_MAX]; };
template <typename T>
__global__ void __launch_bounds__(256)
fd_pack_train_w_h16(const fd_pack_table<T> table)
{
    const fd_pack_rec<T> &R = table.rec[blockIdx.y];
    const float *__restrict__ w = R.w;
    T *__restrict__ wt = R.wt, *__restrict__ wtt = R.wtt;
    const int N = R.N, K = R.K, K64 = R.K64, N64 = R.N64;
    const long a = (long)N * K64, total = a + (long)K * N64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        if (i < a) {
            const int n = (int)(i / K64), k = (int)(i - (long)n * K64);
            fd_st1(wt + i, k < K ? w[(long)n * K + k] : 0.0f);
        } else {
            const long j = i - a;
            const int k = (int)(j / N64), n = (int)(j - (long)k * N64);
            fd_st1(wtt + j, n < N ? w[(long)n * K + k] : 0.0f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Forward.  Skeleton of fd_pw_gemm_h16 (LDS-DMA 3-stage ring, swizzled 128-byte rows, XCD-aware 1-D grid); the scale/shift
// table of the producer sits in LDS as [2][K64] floats, zero beyond K (so a ragged last K tile contributes act(0) * 0).
// Epilogue: z rounded to T, transposed through LDS for 16-byte NHWC stores, per-column statistics of the ROUNDED values
// added to the unit's statistics rows.  fin.rows != null: the producer's BatchNorm is finalised here (fd_stat_table_all after the first LDS-DMA
// stages have been issued; workgroup 0 is the writer).
// ------------------------------------------------------------------------------------------------
template <typename T, int ACT1, int TN = 1>   // TN = column tiles of 32 per wave: the workgroup's tile is 64 x (64*TN).  TN = 2 (N >= 128): every
__global__ void __launch_bounds__(256)        // normalised A fragment feeds two MFMAs (the fp32 table math per fragment is this kernel's VALU load)
fd_pw_gemm_train_h16(const T *__restrict__ A, const float *__restrict__ st1, const T *__restrict__ Wt, T *__restrict__ out,
                     fd_stat_rows sr, int M, int N, int K, int K64, int m_tiles, int n_tiles, fd_bn_fin fin)
{
    constexpr int BM = 64, BN = 64 * TN, BK = 64;
    constexpr int ROWS = BM + BN, STAGE = ROWS * 128, RG = ROWS / 8 / 4;
    FD_DYN_SMEM(smem);
    float *tab = reinterpret_cast<float *>(smem + (K64 < FD_H16_STAGES * BK ? K64 / BK : FD_H16_STAGES) * STAGE);     // [2][K64], behind the ring's min(FD_H16_STAGES, K tiles) stages
    float *red = tab + 2 * K64;                                   // [2][2][BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % n_tiles, mt = (slot / n_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    // the producer's table is requested first and lands in LDS after the first LDS-DMA stages have been issued: one round trip instead of two
    // at the head of every workgroup (the units with K <= 128 run a single K tile: their workgroups are all head and tail)
    constexpr int TABQ = 4;                                       // K64 <= 1024 (checked by the plan)
    float tsv[TABQ], ttv[TABQ];
    if (!fin.rows) {
#pragma unroll
        for (int i = 0; i < TABQ; ++i) {
            const int k = tid + 256 * i;
            const int kc = k < K ? k : 0;
            const float a = st1[FD_ST_SCALE * K + kc], b = st1[FD_ST_SHIFT * K + kc];
            tsv[i] = k < K ? a : 0.0f; ttv[i] = k < K ? b : 0.0f;
        }
    }
    const T *src[RG];
    int src_k[RG];
    bool src_is_a[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const int r = (wave + 4 * i) * 8 + (lane >> 3);
        src_k[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        src_is_a[i] = r < BM;
        if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src[i] = A + row * K; }
        else { int row = n0 + (r - BM); if (row > N - 1) row = N - 1; src[i] = Wt + (long)row * K64; }
    }
    auto issue = [&](int t) {
        unsigned char *dst = smem + (t % FD_H16_STAGES) * STAGE + wave * 8 * 128;
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int k = t * BK + src_k[i];
            if (src_is_a[i] && k >= K) k = 0;
            fd_glds16(reinterpret_cast<const float *>(src[i] + k), reinterpret_cast<float *>(dst + i * 4 * 8 * 128));
        }
    };
    fd_f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const int h = lane >> 5;
    int a_off[4], b_off[TN][4];
    {
        const int ra = wm * 32 + (lane & 31);
#pragma unroll
        for (int s = 0; s < 4; ++s) a_off[s] = ra * 128 + (((2 * s + h) ^ ((ra >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rb = BM + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int s = 0; s < 4; ++s) b_off[j][s] = rb * 128 + (((2 * s + h) ^ ((rb >> 1) & 7)) << 4);
        }
    }
    const int Tn = K64 / BK;
    issue(0);
    if (FD_H16_STAGES > 2 && Tn > 1) issue(1);
    if (fin.rows) fd_stat_table_all<256>(fin, K, K64, tid, blockIdx.x == 0, [&](int k, float a, float b) { tab[k] = a; tab[K64 + k] = b; });
    else {
#pragma unroll
        for (int i = 0; i < TABQ; ++i) {
            const int k = tid + 256 * i;
            if (k < K64) { tab[k] = tsv[i]; tab[K64 + k] = ttv[i]; }
        }
    }
    fd_block_barrier_lds();                                       // table visible
    for (int t = 0; t < Tn; ++t) {
        if (FD_H16_STAGES > 2 && t + 1 < Tn) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();
        fd_block_barrier();
        if (t + FD_H16_STAGES - 1 < Tn) issue(t + FD_H16_STAGES - 1);
        const unsigned char *cur = smem + (t % FD_H16_STAGES) * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kb = t * BK + (2 * s + h) * 8;
            float f[8];
            fd_unpack8(T{}, fd_ld8(cur + a_off[s]), f);
            const fd_f32x4 s0 = fd_ld4(tab + kb), s1 = fd_ld4(tab + kb + 4), t0 = fd_ld4(tab + K64 + kb), t1 = fd_ld4(tab + K64 + kb + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[j] = fd_act<ACT1>(fmaf(f[j], s0[j], t0[j]));
                f[4 + j] = fd_act<ACT1>(fmaf(f[4 + j], s1[j], t1[j]));
            }
            const fd_u16x8 af = fd_pack8(T{}, f);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = fd_mfma_32x32x16(T{}, af, fd_ld8(cur + b_off[j][s]), acc[j]);
        }
    }
    // epilogue, one 32-column tile of the wave at a time
    __syncthreads();
    T *tile = reinterpret_cast<T *>(smem) + wave * 32 * 40;
    const int col = lane & 31;
    const long rbase = m0 + wm * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cw = (wn * TN + j) * 32;                        // first column of this tile within the workgroup's BN columns
        float s = 0.0f, q = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2);
            fd_st1(tile + (rl + 4 * (lane >> 5)) * 40 + col, acc[j][r]);
            const float zr = fd_ld1(tile + (rl + 4 * (lane >> 5)) * 40 + col);          // the rounded value (own write)
            if (rbase + rl < M && n0 + cw + col < N) { s += zr; q = fmaf(zr, zr, q); }
        }
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        if (lane < 32) { red[(wm * 2 + 0) * BN + cw + lane] = s; red[(wm * 2 + 1) * BN + cw + lane] = q; }
        fd_wave_lds_fence();                                      // the tile is private to this wave
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = lane + 64 * i;
            const int row = id >> 2, c8 = (id & 3) * 8;
            const long grow = m0 + wm * 32 + row;
            const int gcol = n0 + cw + c8;
            if (grow < M && gcol < N) fd_st8(out + grow * N + gcol, fd_ld8(tile + row * 40 + c8));   // N % 8 == 0 (checked by the plan)
        }
        fd_wave_lds_fence();
    }
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
        fd_stat_add<FD_STAT_FWD>(sr, mt, N, 0, n0 + tid, red[0 * BN + tid] + red[2 * BN + tid]);
        fd_stat_add<FD_STAT_FWD>(sr, mt, N, 1, n0 + tid, red[1 * BN + tid] + red[3 * BN + tid]);
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward as an elementwise map:  DZ[m][n] = A*((G - C1) - (z - MU)*C2), 8 channels per item.  DZ == G (in place)
// in product plans; plans created with FD_PLAN_KEEP_ACTIVATIONS keep G and write dz to a buffer of its own.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
fd_bn_bwd_apply_h16(const T *G, T *DZ, const T *__restrict__ Z, const float *__restrict__ coef, long chunks, int N)
{
    const int n8 = N >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long)gridDim.x * 256) {
        const int n = (int)(i % n8) * 8;
        float g[8], z[8];
        fd_unpack8(T{}, fd_ld8(G + i * 8), g);
        fd_unpack8(T{}, fd_ld8(Z + i * 8), z);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            g[j] = fd_dz(g[j], z[j], coef[FD_CF_A * N + n + j], coef[FD_CF_C1 * N + n + j], coef[FD_CF_MU * N + n + j], coef[FD_CF_C2 * N + n + j]);
        fd_st8(DZ + i * 8, fd_pack8(T{}, g));
    }
}

// The same map with the unit's BatchNorm-backward finalisation inside (fd_bstat_table_block over the unit's statistics rows): workgroup (x, y) owns channels [64x, 64x + 64) of the rows 32y + r, 32(y + gridDim.y) + r, ...
// (a row's 64 channels = 128 bytes = one line: 8 work-items), sums the partial rows of its channels while its first row's loads are in flight, and
// row y == 0 also writes dgamma / dbeta and the coefficient table.  One launch per unit instead of fd_bn_bwd_finalize_f32 + fd_bn_bwd_apply_h16.
template <typename T>
__global__ void __launch_bounds__(256)
fd_bn_bwd_apply_fin_h16(const T *G, T *DZ, const T *__restrict__ Z, int M, int N, const fd_bn_bwd_fin fin)
{
    __shared__ double sh[512];
    __shared__ float s_cf[4 * 64];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, cl = (tid & 7) * 8;
    const bool c_ok = c0 + cl < N;                         // N % 8 == 0 (16-bit plans): a work-item's 8 channels are all inside or all outside
    const long step = (long)gridDim.y * 32;
    long r = (long)blockIdx.y * 32 + (tid >> 3);
    const long rq = r < M ? r : M - 1;
    const long cq = c_ok ? c0 + cl : 0;
    fd_u16x8 gq = fd_ld8(G + rq * N + cq), zq = fd_ld8(Z + rq * N + cq);
    fd_bstat_table_block(fin, sh, s_cf, c0, 64, N, tid, blockIdx.y == 0);
    float cA[8], c1[8], cM[8], c2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cA[j] = s_cf[FD_CF_A * 64 + cl + j]; c1[j] = s_cf[FD_CF_C1 * 64 + cl + j]; cM[j] = s_cf[FD_CF_MU * 64 + cl + j]; c2[j] = s_cf[FD_CF_C2 * 64 + cl + j]; }
    if (!c_ok) return;
    for (; r < M; r += step) {
        const long rn = r + step < M ? r + step : r;       // the next row of this work-item, requested before this one is formed
        const fd_u16x8 gn = fd_ld8(G + rn * N + cq), zn = fd_ld8(Z + rn * N + cq);
        float g[8], z[8];
        fd_unpack8(T{}, gq, g);
        fd_unpack8(T{}, zq, z);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = fd_dz(g[j], z[j], cA[j], c1[j], cM[j], c2[j]);
        fd_st8(DZ + r * N + cq, fd_pack8(T{}, g));
        gq = gn; zq = zn;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward-data:  G_in[M][K] = mask_in(y_in) * (dz[M][N] x W[N][K] (+ skipgrad)),  + the producer's BN partials.
// Main loop = fd_pw_gemm_h16 with A = dz (row pitch N), B = wtt[K][N64].  Epilogue: the fp32 accumulators are transposed
// through LDS so that z_in / skipgrad are read and G_in is written 8 channels (16 bytes) per lane.
// measurement aid (tools/pw_bwd_phases.py, built with -DFD_PW_PROBE; nothing in product builds): 100 MHz real-time stamps of a workgroup's phases in the
// paired 16-bit pointwise backward launch, for the unit whose (M, N, K) was selected.  Slot b = blockIdx.x: [0] role (1 backward-data, 2 weight-gradient),
// [1..5] stamps, [6] / [7] ticks accumulated in the weight-gradient loop before / after its second barrier (staging incl. the wait for the loads / MFMAs)
#ifdef FD_PW_PROBE
#define FD_PW_PROBE_SLOTS 16384
__device__ long long fd_pw_probe[8 * FD_PW_PROBE_SLOTS];
__device__ int fd_pw_probe_sel[3];
#define FD_PW_PROBE_ON(M_, N_, K_) (threadIdx.x == 0 && blockIdx.x < FD_PW_PROBE_SLOTS && (M_) == fd_pw_probe_sel[0] && (N_) == fd_pw_probe_sel[1] && (K_) == fd_pw_probe_sel[2])
#define FD_PW_PROBE_AT(on, k) do { if (on) fd_pw_probe[8 * blockIdx.x + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define FD_PW_PROBE_SET(on, k, v) do { if (on) fd_pw_probe[8 * blockIdx.x + (k)] = (long long)(v); } while (0)
#define FD_PW_PROBE_NOW() ((long long)__builtin_amdgcn_s_memrealtime())
#else
#define FD_PW_PROBE_ON(M_, N_, K_) false
#define FD_PW_PROBE_AT(on, k) ((void)0)
#define FD_PW_PROBE_SET(on, k, v) ((void)0)
#define FD_PW_PROBE_NOW() 0LL
#endif

// ------------------------------------------------------------------------------------------------
// bytes of the LDS-DMA ring: min(FD_H16_STAGES, N tiles) stages of (64 + 64*TN) 128-byte rows, at least the epilogue's four fp32 [32][36] tiles
#define FD_PW_DGRAD_H16_RING(N64_, TN_) ((size_t)(((N64_) < 64 * FD_H16_STAGES ? (N64_) / 64 : FD_H16_STAGES) * (64 + 64 * (TN_)) * 128 > 4 * 32 * 36 * 4 ? ((N64_) < 64 * FD_H16_STAGES ? (N64_) / 64 : FD_H16_STAGES) * (64 + 64 * (TN_)) * 128 : 4 * 32 * 36 * 4))
template <typename T, int ACT_IN, int ADD_SG, int TN>   // TN: 32-column tiles per wave (workgroup tile 64 x 64*TN of G_in): every dz fragment feeds TN MFMAs
__device__ __forceinline__ void                       // blk: linear workgroup number (blockIdx.x of the plain kernel; the paired launch fd_pw_bwd_h16 passes its own)
fd_pw_dgrad_h16_body(const T *__restrict__ DZ, const T *__restrict__ Wtt, const T *__restrict__ Zin, const float *__restrict__ st_in,
                const T *__restrict__ SG, T *__restrict__ Gin, fd_stat_rows sr, int M, int N, int K, int N64, int m_tiles, int k_tiles,
                const unsigned blk)
{
    constexpr int BM = 64, BKO = 64 * TN, BR = 64;
    constexpr int ROWS = BM + BKO, STAGE = ROWS * 128, RG = ROWS / 8 / 4;
    FD_DYN_SMEM(smem);
    float *red = reinterpret_cast<float *>(smem + FD_PW_DGRAD_H16_RING(N64, TN));     // [2][2][BKO]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wk = wave & 1;
    const int xcd = blk & 7, slot = blk >> 3;
    const int kt = slot % k_tiles, mt = (slot / k_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * BM;
    const int k0 = kt * BKO;
    const bool probe = FD_PW_PROBE_ON(M, N, K);
    FD_PW_PROBE_SET(probe, 0, 1); FD_PW_PROBE_AT(probe, 1);
    const T *src[RG];
    int src_n[RG];
    bool src_is_a[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const int r = (wave + 4 * i) * 8 + (lane >> 3);
        src_n[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        src_is_a[i] = r < BM;
        if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src[i] = DZ + row * N; }
        else { int row = k0 + (r - BM); if (row > K - 1) row = K - 1; src[i] = Wtt + (long)row * N64; }
    }
    auto issue = [&](int t) {
        unsigned char *dst = smem + (t % FD_H16_STAGES) * STAGE + wave * 8 * 128;
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int n = t * BR + src_n[i];
            if (src_is_a[i] && n >= N) n = 0;                     // finite data; the zero-padded weights annihilate it
            fd_glds16(reinterpret_cast<const float *>(src[i] + n), reinterpret_cast<float *>(dst + i * 4 * 8 * 128));
        }
    };
    fd_f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const int h = lane >> 5;
    int a_off[4], b_off[TN][4];
    {
        const int ra = wm * 32 + (lane & 31);
#pragma unroll
        for (int s = 0; s < 4; ++s) a_off[s] = ra * 128 + (((2 * s + h) ^ ((ra >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rb = BM + (wk * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int s = 0; s < 4; ++s) b_off[j][s] = rb * 128 + (((2 * s + h) ^ ((rb >> 1) & 7)) << 4);
        }
    }
    const int Tn = N64 / BR;
    // the epilogue operands of the wave's first column tile (producer's table, z_in, skip gradient) are requested BEFORE the main loop: their
    // round trip runs under it (they are older than every LDS-DMA load, so the loop's vmcnt waits still cover exactly the stages)
    const int c8 = (lane & 3) * 8;
    float psc[8], psh[8], pmu[8], pis[8];
    fd_u16x8 pz[2], pg[2];
    {
        const int gcol0 = k0 + wk * TN * 32 + c8;
        const int gc = gcol0 < K ? gcol0 : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            psc[j] = st_in[FD_ST_SCALE * K + gc + j]; psh[j] = st_in[FD_ST_SHIFT * K + gc + j];
            pmu[j] = st_in[FD_ST_MEAN * K + gc + j]; pis[j] = st_in[FD_ST_INVSTD * K + gc + j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            long grow = m0 + wm * 32 + (lane >> 2) + 16 * i;
            if (grow > M - 1) grow = M - 1;
            pz[i] = fd_ld8(Zin + grow * K + gc);
            if (ADD_SG) pg[i] = fd_ld8(SG + grow * K + gc);
        }
    }
    issue(0);
    if (FD_H16_STAGES > 2 && Tn > 1) issue(1);
    FD_PW_PROBE_AT(probe, 2);
    for (int t = 0; t < Tn; ++t) {
        if (FD_H16_STAGES > 2 && t + 1 < Tn) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();
        fd_block_barrier();
        if (t + FD_H16_STAGES - 1 < Tn) issue(t + FD_H16_STAGES - 1);
        const unsigned char *cur = smem + (t % FD_H16_STAGES) * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const fd_u16x8 af = fd_ld8(cur + a_off[s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = fd_mfma_32x32x16(T{}, af, fd_ld8(cur + b_off[j][s]), acc[j]);
        }
    }
    // epilogue, one 32-column tile of the wave at a time: fp32 tile [32][36] per wave
    FD_PW_PROBE_AT(probe, 3);
    __syncthreads();
    float *tile = reinterpret_cast<float *>(smem) + wave * 32 * 36;
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) {
        const int cw = (wk * TN + jt) * 32;                       // first column of this tile within the workgroup's BKO columns
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[jt][r];
        fd_wave_lds_fence();                                      // the tile is private to this wave
        const int gcol = k0 + cw + c8;
        float sg_[8], sx_[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sg_[j] = 0.0f; sx_[j] = 0.0f; }
        if (gcol < K) {                                           // K % 8 == 0: a chunk is entirely inside or outside
            float sc[8], sh[8], mu[8], is[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (jt == 0) { sc[j] = psc[j]; sh[j] = psh[j]; mu[j] = pmu[j]; is[j] = pis[j]; }
                else {
                    sc[j] = st_in[FD_ST_SCALE * K + gcol + j]; sh[j] = st_in[FD_ST_SHIFT * K + gcol + j];
                    mu[j] = st_in[FD_ST_MEAN * K + gcol + j]; is[j] = st_in[FD_ST_INVSTD * K + gcol + j];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (lane >> 2) + 16 * i;
                const long grow = m0 + wm * 32 + row;
                if (grow < M) {
                    float z[8], v[8];
                    fd_unpack8(T{}, jt == 0 ? pz[i] : fd_ld8(Zin + grow * K + gcol), z);
                    const fd_f32x4 v0 = fd_ld4(tile + row * 36 + c8), v1 = fd_ld4(tile + row * 36 + c8 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
                    if (ADD_SG) {
                        float g[8];
                        fd_unpack8(T{}, jt == 0 ? pg[i] : fd_ld8(SG + grow * K + gcol), g);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += g[j];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= fd_actmask<ACT_IN>(fmaf(z[j], sc[j], sh[j]));
                    const fd_u16x8 packed = fd_pack8(T{}, v);
                    fd_st8(Gin + grow * K + gcol, packed);
                    fd_unpack8(T{}, packed, v);                   // statistics of the stored (rounded) gradient
#pragma unroll
                    for (int j = 0; j < 8; ++j) { sg_[j] += v[j]; sx_[j] = fmaf(v[j], (z[j] - mu[j]) * is[j], sx_[j]); }
                }
            }
        }
        // lanes that share (lane & 3) hold the same 8 columns for different rows
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sg_[j] = fd_row_stride4_sum(sg_[j]); sx_[j] = fd_row_stride4_sum(sx_[j]);
            for (int m = 16; m < 64; m <<= 1) { sg_[j] += __shfl_xor(sg_[j], m); sx_[j] += __shfl_xor(sx_[j], m); }
        }
        if (lane < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(wm * 2 + 0) * BKO + cw + c8 + j] = sg_[j]; red[(wm * 2 + 1) * BKO + cw + c8 + j] = sx_[j]; }
        }
        fd_wave_lds_fence();                                      // the next tile overwrites the wave's LDS tile
    }
    FD_PW_PROBE_AT(probe, 4);
    __syncthreads();
    if (tid < BKO && k0 + tid < K) {
        fd_stat_add<FD_STAT_BWD>(sr, mt, K, 0, k0 + tid, red[0 * BKO + tid] + red[2 * BKO + tid]);
        fd_stat_add<FD_STAT_BWD>(sr, mt, K, 1, k0 + tid, red[1 * BKO + tid] + red[3 * BKO + tid]);
    }
    FD_PW_PROBE_AT(probe, 5);
}

template <typename T, int ACT_IN, int ADD_SG, int TN = 1>
__global__ void __launch_bounds__(256)
fd_pw_dgrad_h16(const T *__restrict__ DZ, const T *__restrict__ Wtt, const T *__restrict__ Zin, const float *__restrict__ st_in,
                const T *__restrict__ SG, T *__restrict__ Gin, fd_stat_rows sr, int M, int N, int K, int N64, int m_tiles, int k_tiles)
{
    fd_pw_dgrad_h16_body<T, ACT_IN, ADD_SG, TN>(DZ, Wtt, Zin, st_in, SG, Gin, sr, M, N, K, N64, m_tiles, k_tiles, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Backward-weights:  wpart[split][n][k] = sum over the split's pixels m of dz[m][n] * a_in[m][k],  a_in = act_in(z_in*s+t).
// Workgroup = 64 (n) x 64 (k) output tile, 64 pixels per step.  The reduction index m is the ROW index of both operands in memory, the MFMA wants
// it as the fast index of a fragment.  LDS images are stored the way the rows arrive -- dz[64 m][64 n], a[64 m][64 k] in T, 16-byte writes, rows of
// 192 bytes -- and the fragments come out of gfx950's LDS transpose read (ds_read_b64_tr_b16, fd_lds_read_tr16: 4 pixels of one column per read,
// two reads per fragment; with the 192-byte pitch the eight 32-byte row pieces of a 32-lane half fall on distinct banks).  Rounds 1-2 transposed on
// the way INTO LDS with 2-byte writes: 32 ds_write_b16 per work-item and step against 8 MFMAs per wave -- that write phase was the kernel's bound.
// ------------------------------------------------------------------------------------------------
#ifndef FD_PW_WGRAD_H16_BUFS
#define FD_PW_WGRAD_H16_BUFS 1          // LDS image sets of the weight-gradient GEMM: 2 = the next step is staged while this step's MFMAs run (one barrier per step); measured: 512 x 512 units 26.1 vs 26.4 us, short-N units slower (49 KB per workgroup in the paired launch), family 477 vs 466 us
#endif
#define FD_PW_WGRAD_H16_LDS(TN_) ((size_t)FD_PW_WGRAD_H16_BUFS * 64 * (1 + (TN_)) * 192)
template <typename T, int ACT_IN, int TN>   // TN: 64-column k tiles per workgroup (output tile 64 n x 64*TN k): the staged dz tile feeds TN times the MFMAs
__device__ __forceinline__ void             // (bx, by) = (output tile, pixel split): blockIdx of the plain kernel; dynamic LDS: FD_PW_WGRAD_H16_LDS(TN) bytes
fd_pw_wgrad_h16_body(const T *__restrict__ DZ, const T *__restrict__ Zin, const float *__restrict__ st_in, float *__restrict__ wpart,
                int M, int N, int K, int k_tiles, int rows_per_split, const int bx, const int by)
{
    constexpr int BT = 64, BR = 64, PITCH = 192;   // 192-byte rows: the transposing 2-byte writes AND the fragment ds_read_b128s are bank-conflict free (144: 2-way read conflicts, PMC)
    FD_DYN_SMEM(smem_w);
    // LDS image set: dz [BR pixels][PITCH] (64 n columns), then TN input tiles [BR pixels][PITCH] (64 k columns each)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int nt = bx / k_tiles, kt = bx - nt * k_tiles;
    const int n0 = nt * BT, k0 = kt * BT * TN;
    const long mbeg = (long)by * rows_per_split;
    long mend = mbeg + rows_per_split; if (mend > M) mend = M;
    const int Tn = (int)((mend - mbeg + BR - 1) / BR);
    const bool probe = FD_PW_PROBE_ON(M, N, K);
    long long pr_stage = 0, pr_mfma = 0;
    FD_PW_PROBE_SET(probe, 0, 2); FD_PW_PROBE_AT(probe, 1);
    // loader mapping: chunk cc = tid & 7 (8 columns), rows lr and lr + 32 of the 64-pixel step
    const int cc = tid & 7, lr = tid >> 3;
    const int ncol = n0 + cc * 8;
    const bool n_ok = ncol < N;                                   // N, K % 8 == 0
    int kcol[TN];
    bool k_ok[TN];
    float sc[TN][8], sh[TN][8];
#pragma unroll
    for (int q = 0; q < TN; ++q) {
        kcol[q] = k0 + q * BT + cc * 8; k_ok[q] = kcol[q] < K;
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[q][j] = k_ok[q] ? st_in[FD_ST_SCALE * K + kcol[q] + j] : 0.0f; sh[q][j] = k_ok[q] ? st_in[FD_ST_SHIFT * K + kcol[q] + j] : 0.0f; }
    }
    fd_u16x8 rdz[2], rzi[2][TN];
    const fd_u16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long m = mbeg + (long)t * BR + lr + 32 * i;
            const bool ok = m < mend;
            const long mq = ok ? m : mend - 1;                 // branch-free: clamped row / column, unconditional loads, select
            const fd_u16x8 vdz = fd_ld8(DZ + mq * N + (n_ok ? ncol : 0));
            rdz[i] = (ok && n_ok) ? vdz : zero8;
#pragma unroll
            for (int q = 0; q < TN; ++q) {
                const fd_u16x8 vzi = fd_ld8(Zin + mq * K + (k_ok[q] ? kcol[q] : 0));
                rzi[i][q] = (ok && k_ok[q]) ? vzi : zero8;
            }
        }
    };
    constexpr int SET = BR * (1 + TN) * PITCH;                  // bytes of one image set (dz + TN input tiles)
    auto stage = [&](int t) {
        unsigned char *s_dz = smem_w + (FD_PW_WGRAD_H16_BUFS > 1 ? (t & 1) * SET : 0), *s_a = s_dz + BR * PITCH;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ml = lr + 32 * i;                           // pixel (row of the LDS images) within the step
            const long m = mbeg + (long)t * BR + ml;
            fd_st8(s_dz + ml * PITCH + cc * 16, rdz[i]);
#pragma unroll
            for (int q = 0; q < TN; ++q) {
                float a[8];
                fd_unpack8(T{}, rzi[i][q], a);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (m < mend && k_ok[q]) ? fd_act<ACT_IN>(fmaf(a[j], sc[q][j], sh[q][j])) : 0.0f;
                fd_st8(s_a + (q * BR + ml) * PITCH + cc * 16, fd_pack8(T{}, a));
            }
        }
    };
    fd_f32x16 acc[TN];
#pragma unroll
    for (int q = 0; q < TN; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    // fragment = 8 consecutive pixels (reduction index) of one column: two LDS transpose reads (fd_lds_read_tr16) of 4 pixels each.  A lane of a
    // 16-lane group points at row (lane & 15) >> 2 of the 4-pixel block, 8-byte chunk (lane & 3) of the group's 16 columns, and receives column lane & 15
    const int hh = lane >> 5;
    const int tr_row = (lane & 15) >> 2;
    const int tr_col_a = (wn * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;      // byte offset within a row of s_dz
    const int tr_col_b = (wk * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;      // ... of a 64-column k tile of s_a
    // (a second register set -- the loads of step t + 2 in flight during step t -- measured slower: 21.9 vs 21.5 us on the 512 x 512 units, 513 vs
    // 500 us for the paired family: the step is bound by the transposing staging writes and the two barriers, not by the loads)
    if (Tn > 0) load(0);
    if (FD_PW_WGRAD_H16_BUFS > 1 && Tn > 0) { stage(0); __syncthreads(); }
    FD_PW_PROBE_AT(probe, 2);
    for (int t = 0; t < Tn; ++t) {
        const long long pr_t0 = probe ? FD_PW_PROBE_NOW() : 0;
        if (FD_PW_WGRAD_H16_BUFS == 1) {
            __syncthreads();                                      // the previous step's fragment reads are done
            stage(t);
            __syncthreads();
        }
        const long long pr_t1 = probe ? FD_PW_PROBE_NOW() : 0;
        if (t + 1 < Tn) load(t + 1);                              // in flight during the MFMAs
        const unsigned char *s_dz = smem_w + (FD_PW_WGRAD_H16_BUFS > 1 ? (t & 1) * SET : 0), *s_a = s_dz + BR * PITCH;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m0 = 16 * s + 8 * hh + tr_row;                // first pixel row this lane points at
            const fd_u16x4 a0 = fd_lds_read_tr16(s_dz + m0 * PITCH + tr_col_a), a1 = fd_lds_read_tr16(s_dz + (m0 + 4) * PITCH + tr_col_a);
            const fd_u16x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
            for (int q = 0; q < TN; ++q) {
                const fd_u16x4 b0 = fd_lds_read_tr16(s_a + (q * BR + m0) * PITCH + tr_col_b), b1 = fd_lds_read_tr16(s_a + (q * BR + m0 + 4) * PITCH + tr_col_b);
                const fd_u16x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                acc[q] = fd_mfma_32x32x16(T{}, a, b, acc[q]);
            }
        }
        if (FD_PW_WGRAD_H16_BUFS > 1) {                          // the other image set is free (its readers passed the previous barrier): stage the next step
            if (t + 1 < Tn) stage(t + 1);
            __syncthreads();
        }
        if (probe) { const long long pr_t2 = FD_PW_PROBE_NOW(); pr_stage += pr_t1 - pr_t0; pr_mfma += pr_t2 - pr_t1; }
    }
    FD_PW_PROBE_AT(probe, 3);
    FD_PW_PROBE_SET(probe, 6, pr_stage); FD_PW_PROBE_SET(probe, 7, pr_mfma);
    float *o = wpart + (long)by * N * K;
    const int rbn = n0 + wn * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < TN; ++q) {
        const int col = k0 + q * BT + wk * 32 + (lane & 31);
        if (col < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbn + (r & 3) + 8 * (r >> 2);
                if (row < N) o[(long)row * K + col] = acc[q][r];
            }
        }
    }
    FD_PW_PROBE_AT(probe, 4);
}
template <typename T, int ACT_IN, int TN = 1>
__global__ void __launch_bounds__(256)
fd_pw_wgrad_h16(const T *__restrict__ DZ, const T *__restrict__ Zin, const float *__restrict__ st_in, float *__restrict__ wpart,
                int M, int N, int K, int k_tiles, int rows_per_split)
{
    fd_pw_wgrad_h16_body<T, ACT_IN, TN>(DZ, Zin, st_in, wpart, M, N, K, k_tiles, rows_per_split, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// One launch for BOTH backward GEMMs of a pointwise unit (they share the operand dz and are independent of each other; as two launches on one
// stream they serialise, and on the 14x14 / 7x7 maps each is a single round of workgroups bound by its own latency chain).  1-D grid: the
// backward-data workgroups first (their partial rows feed the BatchNorm finalisation that follows on the critical path), then the
// n_w = tiles x splits weight-gradient workgroups -- or, round 6, the weight-gradient workgroups first where they are the fewer (n_w > 0): they live 9-15 us against
// the tiles' 3.5-9, and dispatched last they leave every slot one long workgroup to finish (family 484 -> 462 us).  LDS (dynamic) and registers are those of the larger role.
// ------------------------------------------------------------------------------------------------
template <typename T, int ACT_IN, int ADD_SG, int TN>
__global__ void __launch_bounds__(256)
fd_pw_bwd_h16(const T *__restrict__ DZ, const T *__restrict__ Wtt, const T *__restrict__ Zin, const float *__restrict__ st_in,
              const T *__restrict__ SG, T *__restrict__ Gin, fd_stat_rows sr, float *__restrict__ wpart,
              int M, int N, int K, int N64, int m_tiles, int k_tiles_d, int n_dgrad, int k_tiles_w, int tiles_w, int rows_per_split, int n_w)
{
    // (n_w > 0: the weight-gradient workgroups take the FIRST n_w numbers -- they live 9-15 us against the backward-data tiles' 4-10 (profiles/r06/
    // pw_bwd_h16_phase_table.txt): dispatched last they leave a tail of one long workgroup per slot at the end of the launch)
    const int first_d = n_w > 0 ? n_w : 0;
    if ((int)blockIdx.x >= first_d && (int)blockIdx.x < first_d + n_dgrad) {
        fd_pw_dgrad_h16_body<T, ACT_IN, ADD_SG, TN>(DZ, Wtt, Zin, st_in, SG, Gin, sr, M, N, K, N64, m_tiles, k_tiles_d, blockIdx.x - first_d);
    } else {
        const int b = n_w > 0 ? (int)blockIdx.x : (int)blockIdx.x - n_dgrad;
        const int by = b / tiles_w;
        fd_pw_wgrad_h16_body<T, ACT_IN, 1>(DZ, Zin, st_in, wpart, M, N, K, k_tiles_w, rows_per_split, b - by * tiles_w, by);
    }
}
