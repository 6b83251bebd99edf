// fd_kernels_sk_f32.h -- the fp32 pointwise GEMM of fd_kernels_f32.h with a work decomposition that removes the wave
// quantisation of its plain tiled launch ("data-parallel rounds + stream-K remainder").
//
// Why: at batch 32 the 14x14 layers have M = 6272 = 49 * 128 rows, i.e. 784 output tiles of 64x64 for 768 resident workgroup
// slots (256 CUs x 3): the plain launch runs 3.06 "rounds" -- the last 16 tiles own the machine for a whole extra tile time
// (ceiling 76.6 % of the MFMA rate); the 7x7 layers have 400 tiles for 768 slots (some CUs get two
// workgroups, others one).  Here the grid is exactly the resident capacity P and every workgroup executes
//     R = tiles / P   complete tiles (tile r*P + rank: the data-parallel rounds, epilogue straight from registers), then
//     an equal share of the (tile, K-tile) iterations of the remaining tiles % P tiles                       (stream-K).
// A share may cover the end of one tile and the beginning of the next.  A workgroup that computed only part of a tile's K
// range writes its accumulators to a private scratch slot and bumps the tile's counter; the workgroup that finds the counter
// complete ("last arriver") adds all partials IN RANK ORDER (own included, re-read from scratch: the result does not depend
// on who arrives last -> deterministic), applies bias + activation and stores the tile.  No workgroup ever waits for another.
// Counters return to 0 after use (zeroed once by fd_plan_pack_weights).
//
// STATUS: experimental, opt-in (FD_PLAN_STREAMK).  Measured on MI355X, batch 32 (round 1): with all 512 / 768 workgroups
// resident and their shares equal to within one K tile, the 1.64-GMAC layers take 40.2 us -- exactly what the plain launch
// takes with its "3.06 rounds" (conv7.3 40.0 us, conv13.3 39.9 us; 128x64 and 64x64 tiles alike, ~82 TFLOP/s = 0.52 of the fp32
// MFMA peak).  So the plain kernel is NOT limited by wave quantisation as DESIGN.md section 6 first assumed: lightly loaded CUs
// simply finish their tiles faster, i.e. the limiter is a shared resource or the per-wave instruction stream, not the tail.
// The decomposition also makes a frame's result depend (last bits) on its position in the batch, because the K range of a
// tile is split at rank boundaries; the plain kernel keeps frames bit-independent.  Kept for the next round's analysis.
//
// rank = (blockIdx.x % 8) * (P / 8) + blockIdx.x / 8: workgroup b runs on XCD b % 8, so each XCD owns a contiguous range of
// tiles (N tiles of one M tile adjacent -> the A panel is fetched into that XCD's L2 once), and the partial sums of a tile
// almost always meet inside one XCD; visibility across XCDs: see fd_store_dev / fd_load_dev below.
#pragma once
#include "fd_kernels_f32.h"

// (device-coherent access helpers fd_store_dev / fd_load_dev / fd_atomic_inc / fd_release_wg / fd_acquire_wg: fd_device.h)

template <int WGM, int WGN, int TM, int TN, int ACT>
__global__ void __launch_bounds__(256)
fd_pw_gemm_sk_f32(const float *__restrict__ A, const float *__restrict__ Wt, const float *__restrict__ bias, float *__restrict__ out,
                  int M, int N, int K, int K32, int m_tiles, int n_tiles, int dp_rounds, int sk_base, int sk_rem,
                  float *__restrict__ scratch, int *__restrict__ counters)
{
    constexpr int NW = WGM * WGN;
    static_assert(NW == 4, "four waves per workgroup");
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, BK = 32;
    constexpr int ROWS = BM + BN, STAGE = ROWS * BK, RG = ROWS / 8 / NW;
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int P = gridDim.x;
    const int rank = (blockIdx.x & 7) * (P >> 3) + (blockIdx.x >> 3);
    const int T = K32 / BK;
    // stream-K share of this rank, in (tile, K-tile) units of the remainder region: the first sk_rem ranks take sk_base + 1
    auto sk_begin_of = [&](int q) { return q * sk_base + (q < sk_rem ? q : sk_rem); };
    auto rank_of_unit = [&](int u) { const int cut = sk_rem * (sk_base + 1); return (u < cut || sk_base == 0) ? u / (sk_base + 1) : sk_rem + (u - cut) / sk_base; };
    const int sk_begin = sk_begin_of(rank), sk_end = sk_begin_of(rank + 1);
    int sk_u = sk_begin;

    const int h = lane >> 5;
    int a_off[TM][4], b_off[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) a_off[i][g] = row * BK + (((2 * g + h) ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = BM + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) b_off[j][g] = row * BK + (((2 * g + h) ^ ((row >> 1) & 7)) << 2);
    }

    for (int job = 0;; ++job) {
        int tile, kb, ke;
        bool first_sk_segment = false;
        if (job < dp_rounds) { tile = job * P + rank; kb = 0; ke = T; }
        else {
            if (sk_u >= sk_end) break;
            const int rt = sk_u / T;
            tile = dp_rounds * P + rt;
            kb = sk_u - rt * T;
            ke = kb + (sk_end - sk_u); if (ke > T) ke = T;
            first_sk_segment = sk_u == sk_begin;
            sk_u += ke - kb;
        }
        const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
        const long m0 = (long)mt * BM;
        const int n0 = nt * BN;
        if (job > 0) __syncthreads();                        // every wave is done with the previous job's LDS stages

        const float *src[RG];
        int src_chunk[RG];
        bool src_is_a[RG];
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            const int r = (wave + NW * i) * 8 + (lane >> 3);
            src_chunk[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 4;
            src_is_a[i] = r < BM;
            if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src[i] = A + row * K; }
            else { int row = n0 + (r - BM); if (row > N - 1) row = N - 1; src[i] = Wt + (long)row * K32; }
        }
        auto issue = [&](int t) {
            float *dst = smem + ((t - kb) % 3) * STAGE + wave * 8 * BK;
#pragma unroll
            for (int i = 0; i < RG; ++i) {
                int k = t * BK + src_chunk[i];
                if (src_is_a[i] && k >= K) k = 0;
                fd_glds16(src[i] + k, dst + i * NW * 8 * BK);
            }
        };
        float bv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            bv[j] = col < N ? bias[col] : 0.0f;
        }
        fd_f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        issue(kb);
        if (kb + 1 < ke) issue(kb + 1);
        for (int t = kb; t < ke; ++t) {
            if (t + 1 < ke) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();
            fd_block_barrier();
            const float *cur = smem + ((t - kb) % 3) * STAGE;
            fd_f32x4 a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = fd_ld4(cur + a_off[i][0]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = fd_ld4(cur + b_off[j][0]);
            if (t + 2 < ke) issue(t + 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[(g + 1) & 1][i] = fd_ld4(cur + a_off[i][g + 1]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[(g + 1) & 1][j] = fd_ld4(cur + b_off[j][g + 1]);
                }
                FD_SCHED_FENCE();
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][q], b[g & 1][j][q], acc[i][j], 0, 0, 0);
                FD_SCHED_FENCE();
            }
        }

        if (!(kb == 0 && ke == T)) {
            // partial K range: publish, and finish the tile if this workgroup is the last of its contributors
            const int rt = tile - dp_rounds * P;
            const int q_first = rank_of_unit(rt * T), q_last = rank_of_unit(rt * T + T - 1);
            float *mine = scratch + ((long)rank * 2 + (first_sk_segment ? 0 : 1)) * (BM * BN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) fd_store_dev(mine + ((i * TN + j) * 16 + r) * 256 + tid, acc[i][j][r]);
            fd_release_wg();                                  // this wave's partial stores have completed ...
            __syncthreads();                                  // ... and so have every other wave's, before the counter moves
            if (tid == 0) {
                const int old = fd_atomic_inc(counters + rt);
                s_last = old == q_last - q_first;
                if (s_last) fd_store_dev(counters + rt, 0);   // self-cleaning: the next launch that uses this counter finds 0
            }
            __syncthreads();
            if (!s_last) continue;
            fd_acquire_wg();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            for (int q = q_first; q <= q_last; ++q) {
                const float *p = scratch + ((long)q * 2 + (sk_begin_of(q) >= rt * T ? 0 : 1)) * (BM * BN);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += fd_load_dev(p + ((i * TN + j) * 16 + r) * 256 + tid);
            }
        }
        // epilogue (complete tile in registers): bias + activation + store
        const bool full = m0 + BM <= M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
            if (col >= N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const long rbase = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
                float *o = out + rbase * N + col;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * N] = fd_act<ACT>(acc[i][j][r] + bv[j]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (rbase + (r & 3) + 8 * (r >> 2) < M) o[((r & 3) + 8 * (r >> 2)) * N] = fd_act<ACT>(acc[i][j][r] + bv[j]);
                }
            }
        }
    }
}
