// rs_predict.hip -- R-stream replay of draw_f (src/draw-f.cpp:64-73 under R's single stream), PREDICT + VERIFY.
//
// What makes the replay item-sequential is ONE integer per item: where item j's normals start in R's stream is where
// item j - 1's slice loop stopped consuming (src/draw-f.cpp:26,56), i.e. start(j) = start(j - 1) + 2n + 2 + (its
// rejection count).  Everything else -- nu_j = L z_j, the slice loop, f' -- is independent across items ONCE the starts are
// known.  So the replay runs in two phases:
//
//   A  PREDICT   the chain of rng_ess.hip (three items per pass over L, candidates as strided windows of one array of
//                normals), but on a SINGLE-PRECISION copy of L and with nothing written but the predicted counts and start
//                positions: half the bytes per pass (and they fit the 256 MiB Infinity Cache: 134 MB at n = 8192),
//                fp32 MFMA at twice the fp64 rate.  The likelihoods of the trial points are still summed in fp64 from the
//                formula as written; only nu carries fp32's ~1e-6.  A rejection count only changes when a trial point's
//                log-likelihood lies within that error of the slice level: measured ~1e-5 per trial point.
//   B  VERIFY    with the predicted starts every item's normals are known, so nu = L Z is ONE triangular fp64 MFMA
//                product over all columns (the product the item-keyed contract runs, 0.84 of the fp64 MFMA peak) and the
//                slice loops run for all items side by side, in full fp64, the formula as written, on R's uniforms at
//                each item's own position.  Each loop reports what it consumed; rs_commit_kernel walks the items in order
//                and accepts them up to and INCLUDING the first one whose consumption differs from the prediction (its own
//                start was exact, so its result is exact; the start of the next item is corrected) -- everything behind it
//                is discarded and the host runs both phases again from there.  What is committed is therefore exactly what
//                the item-sequential replay computes, whatever the predictor did; a predictor that stalls or is wrong only
//                costs time.  (tests: gpirt_debug_rs_mispredict makes it wrong on purpose.)
//
// GPIRT_RS_PREDICT=2 keeps the one-phase replay of rng_ess.hip (every pass in fp64), which also remains the path for
// the items a stalled predictor leaves over.
#include "common.h"
#include "kernels.h"
#include "ll_fast.h"
#include <vector>

namespace gpirt {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

// L -> single-precision tiles in the order the predictor's products read them: tile (row group rg of 32 rows, oct ko of 8
// columns) = 1 KiB, lane l = (row rg 32 + (l & 31), columns 8 ko + 4 (l >> 5) .. + 3) as one float4: a wave's step is one
// contiguous kilobyte, its steps follow each other in memory.  Rows and columns past the matrix are zeros; the strict upper
// triangle of L holds zeros (gpirt_sampler_create).  Only the tiles a product reads are written.
// diag_only: just the octs of the 512-column part that holds the row group's diagonal (all the structured pass reads, rs_lr.hip)
__global__ __launch_bounds__(256) void rs32_tile_kernel(const double* __restrict__ L, int64_t n, int64_t ldl, int64_t nk8, float* __restrict__ Lt, int diag_only)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rg = blockIdx.y;
    const int64_t r0 = rg * RS_ROWS;
    const int64_t kall = (r0 + RS_ROWS < n) ? r0 + RS_ROWS : n;
    const int64_t ko_end = (kall + 7) / 8;
    const int64_t row = r0 + (lane & 31);
    const int64_t ko_beg = diag_only ? (r0 / RS3P_KC) * (RS3P_KC / 8) : 0;
    for (int64_t ko = ko_beg + (int64_t)blockIdx.x * 4 + wave; ko < ko_end; ko += (int64_t)gridDim.x * 4) {
        const int64_t k = 8 * ko + 4 * (lane >> 5);
        float4 v;
        v.x = (row < n && k + 0 < n) ? (float)L[row + (k + 0) * ldl] : 0.0f;
        v.y = (row < n && k + 1 < n) ? (float)L[row + (k + 1) * ldl] : 0.0f;
        v.z = (row < n && k + 2 < n) ? (float)L[row + (k + 2) * ldl] : 0.0f;
        v.w = (row < n && k + 3 < n) ? (float)L[row + (k + 3) * ldl] : 0.0f;
        *reinterpret_cast<float4*>(Lt + (rg * nk8 + ko) * 256 + 4 * lane) = v;
    }
}

// The predictor's products: part32[by][c][row] = sum over the part's columns of L[row][k] z_c[k], 32 rows x the pass's 32
// candidate columns per work-group (the units of rs3_unit_table), v_mfma_f32_32x32x2f32 with the CANDIDATES as the
// instruction's rows (A) and L's rows as its columns (B): lane l's B operand is one float of its float4 of L (row l & 31,
// column 8 ko + 4 (l >> 5) + s for the s-th instruction of a step), its A operand the normal of ITS candidate c = l & 31 at
// that column -- from the three windows of Nrm the work-group staged in LDS as floats (rng_ess.hip, rs3_product_full: slot 0
// every other normal from the anchor, slots 1 / 2 consecutive normals).  The accumulator's lanes run along L's rows, so
// the stores are 128-byte row segments.  The four waves take a quarter of the part's columns each and meet in LDS.
// A FULL part (every wave has its 16 steps; 9 of 10 work-groups) issues all sixteen kilobytes of its wave up front,
// unconditionally: the compiler's vmcnt bookkeeping then lets step s start when ITS kilobyte has arrived (a load under a
// run-time condition anywhere in the loop makes it wait for every outstanding load at every step: 44 instead of ~20 us per pass).
// A pass places up to FOUR items: the anchor (its start is known: one candidate) and three more, whose starts depend on the
// counts in front of them -- slot 1 tries PD_W1 consecutive starts (k0 = 0 .. 12), slot 2 PD_W2 (k0 + k1 = lo2 .. lo2 + 10),
// slot 3 PD_W3 (k0 + k1 + k2 = lo3 .. lo3 + 6); lo2 / lo3 follow the last draw's mean count (rs_pred_start_kernel).  On the
// counts of a chain at 8192 x 1024 (tools/k_histogram.py) this layout needs 0.33 passes per item in steady state (mean count
// 6.7) and 0.31 at the start (3.5); 1 + 15 + 16 starts for three items: 0.38 / 0.34, with the third's window moved 0.35 / 0.34.
constexpr int PD_SLOTS = 4;
constexpr int PD_W1 = 13, PD_W2 = 11, PD_W3 = 7;
constexpr int PD_B2 = 1 + PD_W1, PD_B3 = PD_B2 + PD_W2;        // first candidate column of slots 2 / 3 (slot 1: column 1)
static_assert(PD_B3 + PD_W3 == RS3_CAND, "the four slots' candidates are the pass's 32 columns");
constexpr int P32_STEPS = RS3P_KC / 32;
constexpr int P32_WIN = 2 * RS3P_KC + 32;                      // a slot's window of normals: every other one from <= 13 consecutive starts
constexpr int P32_LDS = (RS3P_KC + 3 * P32_WIN > 4 * 1024) ? RS3P_KC + 3 * P32_WIN : 4 * 1024;       // the windows, then the four waves' 32 x 32 sums
// (units of 1024 columns instead of 512 -- half the prologues: 37.5 instead of 30 us per pass, gpurun_out/r7l;)
// (steps of L in flight per wave / work-groups per compute unit: 4 ... 16 / 5 ... 8 all measure 30.1-32.3 us per pass -- the pass is
//  bound by the f32 MFMAs at the clock the chip holds under them (26 us with NO loads of L) and by the 134 MB (25 us with NO MFMAs)
//  alike, gpurun_out/r6f, r6u)
#ifndef P32_RINGF
#define P32_RINGF 12
#endif
#ifndef P32_OCC
#define P32_OCC 5
#endif
template <bool FULL>
__device__ __forceinline__ void rs3p_product_unit(const Rs3Args& a, const uint64_t base, const uint64_t off, const uint64_t lo2, const uint64_t lo3, const int bx, const int by, float* lds, long long* tr)
{
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int64_t n = a.n;
    const int64_t r0 = (int64_t)bx * RS_ROWS;
    const int64_t kall = (r0 + RS_ROWS < n) ? r0 + RS_ROWS : n;
    const int64_t k0 = (int64_t)by * RS3P_KC;
    if (!FULL && k0 >= kall) return;
    const int64_t kend = FULL ? k0 + RS3P_KC : ((k0 + RS3P_KC < kall) ? k0 + RS3P_KC : kall);          // this part's columns [k0, kend)
    const int kw = (int)(kend - k0);
    // this wave's steps: octs of 8 columns, a quarter of RS3P_KC per wave
    const int64_t o_beg = (k0 >> 3) + (int64_t)kq * P32_STEPS;
    int steps = P32_STEPS;
    if (!FULL) {
        int64_t o_end = o_beg + P32_STEPS;
        const int64_t o_all = (kend + 7) >> 3;
        if (o_end > o_all) o_end = o_all;
        steps = o_end > o_beg ? (int)(o_end - o_beg) : 0;
    }
    const float* Lp = a.Lt32 + ((int64_t)bx * a.nk8 + o_beg) * 256 + 4 * lane;
    float* W0 = lds; float* W1 = lds + RS3P_KC; float* W2 = W1 + P32_WIN; float* W3 = W2 + P32_WIN;
    static_assert(RS3P_KC + 3 * P32_WIN <= P32_LDS, "the windows must fit");
    static_assert(PD_W1 <= 32 && PD_W2 <= 32 && PD_W3 <= 32, "a window holds 2 RS3P_KC + 32 normals");
    const uint64_t item_step = 2ull * (uint64_t)n + 2ull;
    const double* N0 = a.Nrm + base + 2ull * (uint64_t)k0;
    const double* N1 = N0 + item_step + off;                    // (off: what the anchor's lost rounds have consumed, rs3p_decide_kernel)
    const double* N2 = N1 + item_step + lo2;                    // (lo2 / lo3: where the windows of counts of slots 2 / 3 begin, rs_pred_start_kernel)
    const double* N3 = N1 + 2ull * item_step + lo3;
    float4 lv[P32_RINGF];
    if (FULL) {
        // the windows' loads go out first (L2 hits), the wave's sixteen kilobytes of L straight behind them
        constexpr int C0 = RS3P_KC / 256, CW = (P32_WIN + 255) / 256;
        double w0[C0], w1[CW], w2[CW], w3[CW];
#pragma unroll
        for (int q = 0; q < C0; ++q) w0[q] = N0[2 * (tid + 256 * q)];
#pragma unroll
        for (int q = 0; q < CW; ++q) { const int x = tid + 256 * q; const int xc = x < P32_WIN ? x : 0; w1[q] = N1[xc]; w2[q] = N2[xc]; w3[q] = N3[xc]; }
#pragma unroll
        for (int u = 0; u < P32_RINGF; ++u) lv[u] = *reinterpret_cast<const float4*>(Lp + (int64_t)u * 256);
#pragma unroll
        for (int q = 0; q < C0; ++q) W0[tid + 256 * q] = (float)w0[q];
#pragma unroll
        for (int q = 0; q < CW; ++q) { const int x = tid + 256 * q; if (x < P32_WIN) { W1[x] = (float)w1[q]; W2[x] = (float)w2[q]; W3[x] = (float)w3[q]; } }
    } else {
        // (entries of the windows this part can touch; a ragged last oct reads up to 7 columns past the part -- zeros in the
        // tile, so the normals there must be staged too: an unstaged LDS word may hold a NaN)
        const int kw8 = (kw + 7) & ~7;
        const int x1 = 2 * kw8 + 32;
        for (int x = tid; x < kw8; x += 256) W0[x] = (float)N0[2 * x];
        for (int x = tid; x < x1; x += 256) { W1[x] = (float)N1[x]; W2[x] = (float)N2[x]; W3[x] = (float)N3[x]; }
#pragma unroll
        for (int u = 0; u < P32_RINGF; ++u) lv[u] = (u < steps) ? *reinterpret_cast<const float4*>(Lp + (int64_t)u * 256) : float4{ 0.f, 0.f, 0.f, 0.f };
    }
    __syncthreads();
    if (tr) tr[1] = (long long)wall_clock64();                 // windows staged
    // lane's candidate: column kk of the part is at Zb[zs * kk]
    const float* Zb = (c == 0) ? W0 : (c < PD_B2 ? W1 + (c - 1) : c < PD_B3 ? W2 + (c - PD_B2) : W3 + (c - PD_B3));
    const int zs = (c == 0) ? 1 : 2;
    const float* z = Zb + zs * ((int)((o_beg << 3) - k0) + 4 * hh);
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < P32_STEPS; ++s) {
        if (FULL || s < steps) {                                 // (uniform over the wave)
            const int u = s % P32_RINGF;
            const float4 l = lv[u];
            if (s + P32_RINGF < P32_STEPS) {
                if (FULL) lv[u] = *reinterpret_cast<const float4*>(Lp + (int64_t)(s + P32_RINGF) * 256);   // (static: no run-time condition)
                else lv[u] = (s + P32_RINGF < steps) ? *reinterpret_cast<const float4*>(Lp + (int64_t)(s + P32_RINGF) * 256) : float4{ 0.f, 0.f, 0.f, 0.f };
            }
            const float z0 = z[zs * (8 * s)], z1 = z[zs * (8 * s + 1)], z2 = z[zs * (8 * s + 2)], z3 = z[zs * (8 * s + 3)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z0, l.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z1, l.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z2, l.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z3, l.w, acc, 0, 0, 0);
        }
    }
    // acc[4 q + r] = candidate 8 q + 4 hh + r, row r0 + c.  The quarters meet in LDS: red[kq][cand][row]
    if (tr) tr[2] = (long long)wall_clock64();                 // this wave's MFMAs issued
    __syncthreads();                                            // (the windows are no longer read)
    if (tr) tr[3] = (long long)wall_clock64();
    float* red = lds;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[kq * 1024 + (8 * q + 4 * hh + r) * 32 + c] = acc[4 * q + r];
    __syncthreads();
    float* out = a.part32 + ((int64_t)by * RS3_CAND) * n + r0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int x = tid + 256 * p, cand = x >> 5, row = x & 31;
        const float v = ((red[x] + red[1024 + x]) + red[2048 + x]) + red[3072 + x];
        if (r0 + row < n) out[(int64_t)cand * n + row] = v;
    }
    if (tr) { tr[4] = (long long)wall_clock64(); tr[6] = (long long)clock64(); }
}

__global__ __launch_bounds__(256, P32_OCC) void rs3p_products_kernel(Rs3Args a)
{
    // windows: W0 RS3P_KC floats | W1, W2, W3 2 RS3P_KC + 32 each; then the four waves' 32 x 32 accumulators (16 KB)
    __shared__ __attribute__((aligned(16))) float lds[P32_LDS];
    const uint64_t item0 = a.anchor[0], base = a.anchor[1], stalled = a.anchor[3], off = a.anchor[5], lo2 = a.anchor[6], lo3 = a.anchor[7];
    const uint32_t unit = a.units[blockIdx.x];
    if (item0 >= (uint64_t)a.m || stalled != 0) return;       // every item is predicted, or the predictor has stalled
    const int bx = (int)(unit & 0xffffu), by = (int)(unit >> 16);
    // debug stamps (gpirt_debug_rs_trace; tools/rs_trace.py): first / middle / last full unit, 8 words each from trace[64]
    long long* tr = (a.trace && threadIdx.x == 0 && (blockIdx.x == 0 || (int)blockIdx.x == a.nfull / 2 || (int)blockIdx.x == a.nfull - 1))
                        ? a.trace + 64 + 8 * (blockIdx.x == 0 ? 0 : (int)blockIdx.x == a.nfull / 2 ? 1 : 2) : nullptr;
    if (tr) { tr[0] = (long long)wall_clock64(); tr[5] = (long long)clock64(); }
    if ((int)blockIdx.x < a.nfull) rs3p_product_unit<true>(a, base, off, lo2, lo3, bx, by, lds, tr);
    else                           rs3p_product_unit<false>(a, base, off, lo2, lo3, bx, by, lds, tr);
}

// The structured pass's products (rs_lr.hip: the diagonal parts of L and the coefficient rows C, every unit a whole part of
// RS3P_KC columns): ~300 work-groups, one or two per compute unit, so nothing hides a wave's latencies but the wave itself --
// NW = 8 waves take 64 columns each (8 steps: every kilobyte of the wave's tiles is requested up front) instead of four waves
// 128: the MFMA phase of a unit 3.3 -> 1.8 us (in-kernel stamps), the windows staged by twice the threads.
template <int NW>
__global__ __launch_bounds__(64 * NW) void rs3p_products_lr_kernel(Rs3Args a)
{
    constexpr int NT = 64 * NW, STEPS = RS3P_KC / (8 * NW);
    constexpr int LDSF = (RS3P_KC + 3 * P32_WIN > NW * 1024) ? RS3P_KC + 3 * P32_WIN : NW * 1024;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const uint64_t item0 = a.anchor[0], base = a.anchor[1], stalled = a.anchor[3], off = a.anchor[5], lo2 = a.anchor[6], lo3 = a.anchor[7];
    const uint32_t unit = a.units[blockIdx.x];
    if (item0 >= (uint64_t)a.m || stalled != 0) return;       // every item is predicted, or the predictor has stalled
    const int bx = (int)(unit & 0xffffu), by = (int)((unit >> 16) & 0x7fffu);
    const bool ext = (unit >> 31) != 0;
    long long* tr = (a.trace && threadIdx.x == 0 && (blockIdx.x == 0 || (int)blockIdx.x == a.nfull / 2 || (int)blockIdx.x == a.nfull - 1))
                        ? a.trace + 64 + 8 * (blockIdx.x == 0 ? 0 : (int)blockIdx.x == a.nfull / 2 ? 1 : 2) : nullptr;
    if (tr) { tr[0] = (long long)wall_clock64(); tr[5] = (long long)clock64(); }
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int64_t n = a.n;
    const int64_t r0 = (int64_t)bx * RS_ROWS;
    const int64_t k0 = (int64_t)by * RS3P_KC;
    const int64_t o_beg = (k0 >> 3) + (int64_t)kq * STEPS;
    const float* Lp = (ext ? a.Ct32 : a.Lt32) + ((int64_t)bx * a.nk8 + o_beg) * 256 + 4 * lane;
    float* W0 = lds; float* W1 = lds + RS3P_KC; float* W2 = W1 + P32_WIN; float* W3 = W2 + P32_WIN;
    const uint64_t item_step = 2ull * (uint64_t)n + 2ull;
    const double* N0 = a.Nrm + base + 2ull * (uint64_t)k0;
    const double* N1 = N0 + item_step + off;
    const double* N2 = N1 + item_step + lo2;
    const double* N3 = N1 + 2ull * item_step + lo3;
    constexpr int C0 = (RS3P_KC + NT - 1) / NT, CW = (P32_WIN + NT - 1) / NT;
    double w0[C0], w1[CW], w2[CW], w3[CW];
    float4 lv[STEPS];
#pragma unroll
    for (int q = 0; q < C0; ++q) { const int x = tid + NT * q; w0[q] = N0[2 * (x < RS3P_KC ? x : 0)]; }
#pragma unroll
    for (int q = 0; q < CW; ++q) { const int x = tid + NT * q; const int xc = x < P32_WIN ? x : 0; w1[q] = N1[xc]; w2[q] = N2[xc]; w3[q] = N3[xc]; }
#pragma unroll
    for (int u = 0; u < STEPS; ++u) lv[u] = *reinterpret_cast<const float4*>(Lp + (int64_t)u * 256);
#pragma unroll
    for (int q = 0; q < C0; ++q) { const int x = tid + NT * q; if (x < RS3P_KC) W0[x] = (float)w0[q]; }
#pragma unroll
    for (int q = 0; q < CW; ++q) { const int x = tid + NT * q; if (x < P32_WIN) { W1[x] = (float)w1[q]; W2[x] = (float)w2[q]; W3[x] = (float)w3[q]; } }
    __syncthreads();
    if (tr) tr[1] = (long long)wall_clock64();                 // windows staged
    const float* Zb = (c == 0) ? W0 : (c < PD_B2 ? W1 + (c - 1) : c < PD_B3 ? W2 + (c - PD_B2) : W3 + (c - PD_B3));
    const int zs = (c == 0) ? 1 : 2;
    const float* z = Zb + zs * ((int)((o_beg << 3) - k0) + 4 * hh);
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const float4 l = lv[s];
        const float z0 = z[zs * (8 * s)], z1 = z[zs * (8 * s + 1)], z2 = z[zs * (8 * s + 2)], z3 = z[zs * (8 * s + 3)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z0, l.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z1, l.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z2, l.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z3, l.w, acc, 0, 0, 0);
    }
    if (tr) tr[2] = (long long)wall_clock64();                 // this wave's MFMAs issued
    __syncthreads();                                            // (the windows are no longer read)
    if (tr) tr[3] = (long long)wall_clock64();
    float* red = lds;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[kq * 1024 + (8 * q + 4 * hh + r) * 32 + c] = acc[4 * q + r];
    __syncthreads();
    // the diagonal part's product is part 0 of part32; C's rows go to the part's record of RS_LR_RANK values
    float* out = ext ? a.lrY + ((int64_t)by * RS3_CAND) * RS_LR_RANK + r0 : a.part32 + r0;
    const int64_t os = ext ? RS_LR_RANK : n, lim = os - r0;
#pragma unroll
    for (int p = 0; p < 1024 / NT; ++p) {
        const int x = tid + NT * p, cand = x >> 5, row = x & 31;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w * 1024 + x];
        if (row < lim) out[(int64_t)cand * os + row] = v;
    }
    if (tr) { tr[4] = (long long)wall_clock64(); tr[6] = (long long)clock64(); }
}

// ---- the predictor's slice loops: ONE meeting per pass -------------------------------------------------------------------
// rs3_slice_kernel (rng_ess.hip) runs a pass's three slice loops one after the other, each a round of eight trial points and a
// meeting of 256 work-groups: 31 us per pass, more than the predictor's products.  Nothing here has to follow the reference's
// arithmetic to the letter -- only the COUNTS come out, and the exact phase verifies them -- so the predictor turns the loops
// inside out: the trial points of a slice loop do not depend on the data (src/draw-f.cpp:50-56: a rejected point only moves the
// bracket end of its own sign), hence the first PD_T points of EVERY candidate start of all four items can be evaluated side
// by side before anything is decided.  Work-group (candidate c, row part p) of 32 x PD_PARTS: walks candidate c's bracket
// sequence from R's uniforms at ITS position in the stream, takes cos / sin of the PD_T points, and sums, over its rows,
// ll(f) and the PD_T trial log-likelihoods of the item the candidate belongs to -- the term in single precision through the
// hardware's exp / log (ll_term_screen, ll_fast.h) -- on nu = the candidate's column of the predictor's products.  PD_T + 1
// partial sums per work-group, stored write-through; a ticket; the work-group whose ticket comes last adds the parts in a
// fixed order and runs the three decisions (accept = first point above the slice level, draw-f.cpp:45) in one thread:
// counts, start positions, the next anchor.  A loop that rejects all PD_T points ends the pass there; the item becomes the
// next anchor with round + 1, and a pass evaluates points round * PD_T .. of its anchor item (the products are recomputed:
// one pass in ten at 8192 x 1024).
constexpr int PD_T = 16;                 // trial points per candidate and pass
constexpr int PD_PARTS = 8;              // row parts per candidate
constexpr int PD_V = PD_T + 1;           // sums per work-group: ll(f), then the trial points
constexpr int PD_MAXROUND = 14;          // 2 + PD_T * (PD_MAXROUND + 1) uniforms staged <= 256
constexpr int PD_REC = PD_T + 3;         // per candidate: uniforms consumed when point t is the current one (-1: past the window), log(u), valid,
                                         // uniforms consumed when the NEXT round's first point is the current one

// LR: the structured form (rs_lr.hip) -- a candidate's column is ONE part, completed by rs_lr_apply_kernel
template <bool LR>
__global__ __launch_bounds__(320) void rs3p_decide_kernel(Rs3Args a)
{
    extern __shared__ double pd_pad[];    // (dynamic LDS only to keep these work-groups one per compute unit: the ticket's hand-off was measured that way)
    __shared__ double uL[256];
    __shared__ double rec[PD_REC];
    __shared__ double cs[2 * PD_T];
    __shared__ double tot[RS3_CAND][PD_V];
    __shared__ double recs[RS3_CAND][PD_REC];
    __shared__ unsigned s_old;
    __shared__ int hitv[RS3_CAND], nanv[RS3_CAND];
    const int tid = threadIdx.x;
    const uint64_t item0 = a.anchor[0], start = a.anchor[1], nrm_end = a.anchor[2], stalled = a.anchor[3], round0 = a.anchor[4], off0 = a.anchor[5], lo2 = a.anchor[6], lo3 = a.anchor[7];
    if (item0 >= (uint64_t)a.m || stalled != 0) return;       // every item is predicted, or the predictor has stalled
    const int64_t n = a.n;
    const int c = (int)blockIdx.x / PD_PARTS, p = (int)blockIdx.x % PD_PARTS;
    const int g = (c == 0) ? 0 : (c < PD_B2 ? 1 : c < PD_B3 ? 2 : 3);
    const int64_t j = (int64_t)item0 + g;
    const uint64_t item_step = 2ull * (uint64_t)n + 2ull;
    // (an anchor in round r has consumed off0 uniforms in its lost rounds: the candidate starts of the items behind it begin there)
    const uint64_t start_c = start + (uint64_t)g * item_step + (c == 0 ? 0ull : off0 + (g == 1 ? (uint64_t)(c - 1) : g == 2 ? lo2 + (uint64_t)(c - PD_B2) : lo3 + (uint64_t)(c - PD_B3)));
    const uint64_t p0 = start_c + 2ull * (uint64_t)n;           // behind the n normals
    const int rnd = (c == 0) ? (int)round0 : 0;
    // the item's 2n uniforms (all its normals were built) and its first two slice uniforms lie inside the window
    const bool valid = j < a.m && !(start_c + 2ull * (uint64_t)n + 2ull > a.cap || start_c + 2ull * (uint64_t)(n - 1) >= nrm_end);
    long long* tr = (a.trace && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) ? a.trace + (blockIdx.x == 0 ? 0 : 16) : nullptr;
    int ti = 0;
    auto stamp = [&]() { if (tr && ti < 16) tr[ti] = (long long)wall_clock64(); ++ti; };
    stamp();
    const int nU = 2 + PD_T * (rnd + 1);
    // Five waves: waves 0..3 (tid < 256) are the row workers; wave 4 alone fetches the uniforms, walks the bracket sequence
    // and takes the cos / sin while the others load and add their rows -- they meet at ONE barrier
    const int wv = tid >> 6, lane = tid & 63;
    const bool worker = tid < 256;
    double uval[4];
    if (wv == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = lane + 64 * k;
            const uint64_t q = p0 + (uint64_t)x;
            uval[k] = (x < nU && valid && q < a.cap) ? a.U[q] : __builtin_nan("");
        }
    }
    // this thread's rows: r_beg + tid + 256 e, PD_RB of them at a time
    const int64_t rp = ((n + (int64_t)PD_PARTS * 256 - 1) / ((int64_t)PD_PARTS * 256)) * 256;
    const int64_t r_beg = (int64_t)p * rp;
    const int64_t r_end = (r_beg + rp < n) ? r_beg + rp : n;
    const double* fj = a.f + (valid ? j : 0) * n; const double* mj = a.mu + (valid ? j : 0) * n; const double* yj = a.y + (valid ? j : 0) * n;
    constexpr int PD_RB = 4;
    float yv[PD_RB], fv[PD_RB], mv[PD_RB], nv[PD_RB];
    double yd[PD_RB], fd[PD_RB], md[PD_RB];
    // parts of the candidate's column that can hold something for this work-group's rows: the parts BEHIND a row's own columns
    // were never written by the products and still hold the zeros of the sampler's creation, so every row simply adds pmax parts
    const int pmax = LR ? 1 : (int)((((r_end < n ? r_end : n) + RS3P_KC - 1) / RS3P_KC));
    const int64_t pstep = (int64_t)RS3_CAND * n;
    // Every load of a row is UNCONDITIONAL (rows past the part's end read the last row of the matrix and are masked when the
    // values are used): a load under a predicate is followed by its select, and the select waits for the load -- twelve
    // round trips one after the other instead of one (2.0 of the prologue's 5 us, in-kernel stamps).
    constexpr int PD_QB = LR ? 1 : 16;                         // parts of a row in flight together (n <= 8192: all of them)
    float t16[PD_RB][PD_QB];
    int64_t ro[PD_RB];
    auto issue_rows = [&](const int64_t i0) {
#pragma unroll
        for (int e = 0; e < PD_RB; ++e) ro[e] = (i0 + 256 * e < n) ? i0 + 256 * e : n - 1;
#pragma unroll
        for (int e = 0; e < PD_RB; ++e) { yd[e] = yj[ro[e]]; fd[e] = fj[ro[e]]; md[e] = mj[ro[e]]; }
        const float* pb = a.part32 + (int64_t)c * n;
#pragma unroll
        for (int u = 0; u < PD_QB; ++u) {
            const float* pq = pb + (int64_t)(u < pmax ? u : 0) * pstep;          // (uniform)
#pragma unroll
            for (int e = 0; e < PD_RB; ++e) t16[e][u] = pq[ro[e]];
        }
    };
    auto finish_rows = [&](const int64_t i0) {
        const float* pb = a.part32 + (int64_t)c * n;
#pragma unroll
        for (int e = 0; e < PD_RB; ++e) {
            const bool in = valid && i0 + 256 * e < r_end;
            yv[e] = in ? (float)yd[e] : __builtin_nanf(""); fv[e] = (float)fd[e]; mv[e] = (float)md[e]; nv[e] = 0.0f;
#pragma unroll
            for (int u = 0; u < PD_QB; ++u) nv[e] += (u < pmax) ? t16[e][u] : 0.0f;
        }
        for (int q0 = PD_QB; q0 < pmax; q0 += 4) {             // (n > 8192: the parts behind the sixteenth, four at a time)
            float t4[PD_RB][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* pq = pb + (int64_t)(q0 + u < pmax ? q0 + u : 0) * pstep;       // (uniform)
#pragma unroll
                for (int e = 0; e < PD_RB; ++e) t4[e][u] = pq[ro[e]];
            }
#pragma unroll
            for (int e = 0; e < PD_RB; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u) nv[e] += (q0 + u < pmax) ? t4[e][u] : 0.0f;
        }
    };
    if (worker) {
        issue_rows(r_beg + tid);
        if (tid == 64) rec[PD_T] = (valid && p0 < a.cap) ? log(a.U[p0]) : 0.0;       // log(u), :28-29
        finish_rows(r_beg + tid);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) uL[lane + 64 * k] = uval[k];
        // (one wave: its LDS accesses execute in order; the wait only keeps the compiler from moving them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // The bracket walk (src/draw-f.cpp:33-36, 50-56) from the stream alone, every point as if all before it were rejected
        // -- by EVERY lane of the wave alike (uniform control flow, LDS broadcasts), lane t keeping point t for its cos / sin.
        int uidx = 1; bool bad = false;                              // (uL[0] = u of :28: its logarithm is a worker's)
        auto next_u = [&]() -> double { const double x = uL[uidx]; ++uidx; if (x != x) { bad = true; return 0.5; } return x; };
        double eps_min = 0.0, eps_max = GP_2PI;
        double eps = eps_min + (eps_max - eps_min) * next_u();
        eps_min = eps - GP_2PI;
        const int t_beg = rnd * PD_T;
        for (int t = 0; t < t_beg; ++t) {                            // (rounds this item has already lost)
            if (eps < 0.0) eps_min = eps; else eps_max = eps;
            if (eps_min == eps_max) eps = eps_min; else eps = eps_min + (eps_max - eps_min) * next_u();
        }
        // the round's uniforms read together, as if every rejection consumed one -- true unless the bracket closes
        // (eps_min == eps_max, where :56 consumes nothing): then the round is walked again with dependent reads
        double uu[PD_T];
#pragma unroll
        for (int t = 0; t < PD_T; ++t) uu[t] = uL[uidx + t];
        double my_eps = 0.0, my_rec = 0.0, end_rec = 0.0;
        {
            double e0 = eps, emin = eps_min, emax = eps_max;
            bool closed = false, badf = bad;
#pragma unroll
            for (int t = 0; t < PD_T; ++t) {
                if (lane == t) { my_eps = e0; my_rec = badf ? -1.0 : (double)(uidx + t); }
                if (e0 < 0.0) emin = e0; else emax = e0;            // :50-55
                if (emin == emax) closed = true;
                double x = uu[t];
                if (x != x) { badf = true; x = 0.5; }
                e0 = emin + (emax - emin) * x;                      // :56
            }
            end_rec = badf ? -1.0 : (double)(uidx + PD_T);
            if (closed) {
#pragma unroll 1
                for (int t = 0; t < PD_T; ++t) {
                    if (lane == t) { my_eps = eps; my_rec = bad ? -1.0 : (double)uidx; }
                    if (eps < 0.0) eps_min = eps; else eps_max = eps;
                    if (eps_min == eps_max) eps = eps_min;          // R::runif(a, a) = a, nothing consumed
                    else eps = eps_min + (eps_max - eps_min) * next_u();
                }
                end_rec = bad ? -1.0 : (double)uidx;
            }
        }
        if (lane < PD_T) {
            double sn, cn;
            sincos(my_eps, &sn, &cn);
            cs[lane] = cn; cs[PD_T + lane] = sn; rec[lane] = my_rec;
        }
        if (lane == 0) { rec[PD_T + 1] = valid ? 1.0 : 0.0; rec[PD_T + 2] = end_rec; }
    }
    __syncthreads();
    stamp();
    // the terms: single precision throughout (argument, exp, log, and the sum over this thread's few rows) -- what comes out
    // is a PREDICTION; per row 1e-7 relative, the same order as the single-precision nu
    float acc[PD_V];
#pragma unroll
    for (int v = 0; v < PD_V; ++v) acc[v] = 0.0f;
    {
        float ct[PD_T], st[PD_T];
#pragma unroll
        for (int t = 0; t < PD_T; ++t) { ct[t] = (float)cs[t]; st[t] = (float)cs[PD_T + t]; }
        // log(1 + exp(-a)) = max(-a, 0) + ln 2 * log2(1 + 2^(-|a| log2 e)) on the bare v_exp_f32 / v_log_f32: the argument of the
        // exponential is <= 0 (an underflow is the right answer, 0) and that of the logarithm lies in [1, 2], so the library
        // forms' range checks (22 instructions per term instead of 8) have nothing to catch
        auto term = [](float arg) -> float {
            const float e = __builtin_amdgcn_exp2f(fabsf(arg) * -1.44269504088896341f);
            return __builtin_fmaf(__builtin_amdgcn_logf(1.0f + e), 0.693147180559945309f, fmaxf(-arg, 0.0f));
        };
        for (int64_t i0 = r_beg + tid; worker && i0 < r_end; i0 += 256 * PD_RB) {
            if (i0 != r_beg + tid) { issue_rows(i0); finish_rows(i0); }
#pragma unroll
            for (int e = 0; e < PD_RB; ++e) {
                if (yv[e] == yv[e]) {
                    const float yf = yv[e] * fv[e], yn = yv[e] * nv[e], ym = yv[e] * mv[e];      // (y = +-1: exact)
                    acc[0] += term(yf + ym);
#pragma unroll
                    for (int t = 0; t < PD_T; ++t) acc[1 + t] += term(__builtin_fmaf(yf, ct[t], __builtin_fmaf(yn, st[t], ym)));
                }
            }
        }
    }
    stamp();
    // the work-group's PD_V sums: through LDS (sred[v][thread]), 16 threads per sum, then one thread per sum -- fixed order
    {
        float* sred = reinterpret_cast<float*>(pd_pad);          // PD_V x 256 floats of the dynamic LDS
#pragma unroll
        for (int v = 0; v < PD_V; ++v) if (worker) sred[v * 256 + tid] = acc[v];
        __syncthreads();
        double* sred2 = pd_pad + (PD_V * 256) / 2 + 8;           // PD_V x 16 doubles behind it
        for (int x = tid; x < PD_V * 16; x += 320) {
            const int v = x >> 4, ch = x & 15;
            double r = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) r += (double)sred[v * 256 + 16 * ch + q];
            sred2[x] = r;
        }
        __syncthreads();
        if (tid < PD_V) {
            double r = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) r += sred2[tid * 16 + q];
            __hip_atomic_store(a.dec_part + (size_t)blockIdx.x * PD_V + tid, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (p == 0 && tid >= 64 && tid < 64 + PD_REC)
        __hip_atomic_store(a.dec_rec + (size_t)c * PD_REC + (tid - 64), rec[tid - 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef GPIRT_PANEL_FENCES
    // the fenced reference form of the ticket (`make fences`, tests/test_gpu_fences.py): release in front of the add, acquire
    // behind it in the work-group that arrives last
    if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
    // The ticket, in two levels: 256 adds to ONE word queue up in the L2 (~12 ns each: the last of them waited ~3 us); the
    // work-groups of one row part (32 of them, one per candidate) share a word on a line of its own, and the last of each
    // adds to the top word.  Whoever is last THERE knows every part is in memory: each arrival's stores had been waited for
    // before its add, and each first-level "last" saw all of its group's adds.  (A stale read here could only cost a
    // misprediction -- the exact phase verifies every count.)
    if (tid == 0) {
        unsigned last = 0;
        const unsigned o1 = __hip_atomic_fetch_add(a.dec_ticket + 32 * (1 + p), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((o1 % (unsigned)RS3_CAND) == (unsigned)RS3_CAND - 1u) {
            const unsigned o2 = __hip_atomic_fetch_add(a.dec_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (o2 % (unsigned)PD_PARTS) == (unsigned)PD_PARTS - 1u;
        }
        s_old = last;
    }
    __syncthreads();
    stamp();
    if (s_old == 0) return;                                     // not the last to arrive
#ifdef GPIRT_PANEL_FENCES
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    long long* tr2 = (a.trace && tid == 0) ? a.trace + 32 : nullptr;             // (debug stamps of the LAST ARRIVER: absolute clock)
    if (tr2) { tr2[0] = (long long)wall_clock64(); tr2[4] = blockIdx.x; }
    // ---- the last work-group: every part is in memory (each storing wave waited for its write-through stores before its
    // work-group's ticket)
    // (MI355X_MICROARCH.md, hand-offs measured with sc1 loads in place of the acquire, first row: one lane of each storing
    // work-group adds to ONE counter after every storing wave's vmcnt(0) wait and its barrier; the work-group whose add came
    // last loads after that add has returned and a barrier; stores and loads all 8-byte sc1; hipMalloc memory; one work-group
    // per compute unit -- which the dynamic LDS of this launch enforces)
    {
        // (all of this thread's loads in flight together: a write-through line of another XCD is a trip to memory)
        constexpr int NX = (RS3_CAND * PD_V + 255) / 256, NR = (RS3_CAND * PD_REC + 255) / 256;
        double pv[NX][PD_PARTS], rv[NR];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int x = tid + 256 * k, cc = x / PD_V, v = x % PD_V;
#pragma unroll
            for (int q = 0; q < PD_PARTS; ++q)
                pv[k][q] = (worker && x < RS3_CAND * PD_V) ? __hip_atomic_load(a.dec_part + ((size_t)(cc * PD_PARTS + q)) * PD_V + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int x = tid + 256 * k;
            rv[k] = (worker && x < RS3_CAND * PD_REC) ? __hip_atomic_load(a.dec_rec + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int x = tid + 256 * k;
            double r = 0.0;
#pragma unroll
            for (int q = 0; q < PD_PARTS; ++q) r += pv[k][q];
            if (worker && x < RS3_CAND * PD_V) tot[x / PD_V][x % PD_V] = r;
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int x = tid + 256 * k;
            if (worker && x < RS3_CAND * PD_REC) recs[x / PD_REC][x % PD_REC] = rv[k];
        }
    }
    __syncthreads();
    if (tr2) tr2[1] = (long long)wall_clock64();
    if (tid < RS3_CAND) { hitv[tid] = 99; nanv[tid] = 99; }
    __syncthreads();
    // every (candidate, point) comparison side by side: the first point above the slice level (:45-47), the first NaN
    for (int x = tid; x < RS3_CAND * PD_T; x += 320) {
        const int cc = x / PD_T, t = x % PD_T;
        const double llp = -tot[cc][1 + t];
        const double log_y = -tot[cc][0] + recs[cc][PD_T];                    // draw-f.cpp:28-29
        if (llp > log_y) atomicMin(&hitv[cc], t);
        else if (llp != llp) atomicMin(&nanv[cc], t);
    }
    __syncthreads();
    stamp();
    if (tid != 0) return;
    const int ns = ((int64_t)a.m - (int64_t)item0 < PD_SLOTS) ? (int)((int64_t)a.m - (int64_t)item0) : PD_SLOTS;
    // usum: uniforms the slots in front have consumed beyond their first two -- counted from the anchor's start, so in a pass
    // whose anchor has lost rounds it begins at off0 and the candidate columns of slots 1 / 2 are taken relative to that
    int usum = 0, resolved = 0, stall = 0;
    uint64_t cur = start, next_round = 0, next_off = 0;
    for (int gg = 0; gg < ns; ++gg) {
        int col = 0;
        if (gg >= 1) {
            const int rel = usum - (int)off0 - (gg == 1 ? 0 : gg == 2 ? (int)lo2 : (int)lo3);
            const int wd = (gg == 1) ? (PD_W1 < a.lim1 ? PD_W1 : a.lim1) : ((gg == 2 ? PD_W2 : PD_W3) < a.lim2 ? (gg == 2 ? PD_W2 : PD_W3) : a.lim2);
            if (rel < 0 || rel >= wd) break;
            col = (gg == 1 ? 1 : gg == 2 ? PD_B2 : PD_B3) + rel;
        }
        const int64_t jj = (int64_t)item0 + gg;
        if (recs[col][PD_T + 1] == 0.0) { stall = 1; break; }               // past the window: the exact phase reports it if it is real
        const int rg = (gg == 0) ? (int)round0 : 0;
        const int hv = hitv[col], nn = nanv[col];
        const int hit = hv < PD_T ? hv : -1;
        if (nn < (hit < 0 ? PD_T : hit)) { stall = 1; break; }                 // NaN state: never accepts
        if (hit < 0) {                                                         // all PD_T points rejected: the next pass goes on from here
            next_round = (uint64_t)(rg + 1);
            if (rg + 1 > PD_MAXROUND || recs[col][PD_T + 2] < 2.0) stall = 1;
            else next_off = (uint64_t)((int)recs[col][PD_T + 2] - 2);
            break;
        }
        if (recs[col][hit] < 0.0) { stall = 1; break; }                       // a uniform past the window
        int uacc = (int)recs[col][hit];
        // (tests: gpirt_debug_rs_mispredict makes the predictor wrong on purpose at every mis-th item)
        if (a.mispredict > 0 && (jj % a.mispredict) == a.mispredict - 1) uacc += 1;
        cur = cur + 2ull * (uint64_t)n + (uint64_t)uacc;
        a.k_out[jj] = rg * PD_T + hit;
        a.posv[jj + 1] = cur;
        usum += uacc - 2;
        ++resolved;
    }
    a.anchor[0] = item0 + (uint64_t)resolved;
    a.anchor[1] = cur;
    a.anchor[4] = next_round;
    a.anchor[5] = next_off;
    if (a.pass_count) *a.pass_count += 1;                      // (real passes so far: the host sizes the next draw's launches by it)
    if (stall) a.anchor[3] = 1;
    if (tr2) tr2[2] = (long long)wall_clock64();
    stamp();
    if (tr) a.trace[15 + (blockIdx.x == 0 ? 0 : 16)] = ti;
}

// The predictor starts where the exact state stands.  It also places slot 2's candidate window: the sixteen starts it can try
// cover k0 + k1 in [lo2, lo2 + 16), and the sum of two counts has its mass around twice the mean count (8192 x 1024 in steady
// state: mean 6.7, [0, 16) holds 68 % of the sums, [6, 22) 90 %; tools/k_histogram.py) -- so the window is centred on twice the
// mean of the counts of the last draw.  (Whatever it is, only the number of passes depends on it.)
__global__ __launch_bounds__(256) void rs_pred_start_kernel(const uint64_t* __restrict__ anchor, uint64_t* __restrict__ anchorP,
                                                            const int* __restrict__ k_last, int64_t m)
{
    __shared__ long long part[256];
    long long sum = 0;
    if (k_last)
        for (int64_t j = threadIdx.x; j < m; j += 256) { const int k = k_last[j]; sum += (k < 0) ? 0 : (k > 64 ? 64 : k); }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int q = 1; q < 256; ++q) sum += part[q];
    long long lo2 = 0, lo3 = 0;
    if (k_last && m > 0) {
        const double kbar = (double)sum / (double)m;
        lo2 = (long long)floor(2.0 * kbar - 6.0);
        lo3 = (long long)floor(3.0 * kbar - 4.0);
        lo2 = lo2 < 0 ? 0 : (lo2 > 24 ? 24 : lo2);
        lo3 = lo3 < 0 ? 0 : (lo3 > 40 ? 40 : lo3);
    }
    anchorP[0] = anchor[0]; anchorP[1] = anchor[1]; anchorP[2] = anchor[2]; anchorP[3] = 0; anchorP[4] = 0; anchorP[5] = 0;
    anchorP[6] = (uint64_t)lo2; anchorP[7] = (uint64_t)lo3;
}

// Z[:, j - j0] = the normals of item j at its predicted start, j in [j0, predicted); zeros behind
__global__ __launch_bounds__(256) void rs_gather_kernel(const double* __restrict__ Nrm, const uint64_t* __restrict__ posv,
                                                        const uint64_t* __restrict__ anchorP, int64_t n, int64_t j0, int64_t m,
                                                        double* __restrict__ Z)
{
    const int64_t j = j0 + blockIdx.x;
    const bool have = (uint64_t)j < anchorP[0];
    const uint64_t p = have ? posv[j] : 0;
    for (int64_t i = threadIdx.x; i < n; i += 256)
        Z[(j - j0) * n + i] = have ? Nrm[p + 2ull * (uint64_t)i] : 0.0;
}

// ess() (src/draw-f.cpp:21-60) of item j = j0 + blockIdx.x at its predicted start, in full: the formula as written, R's
// uniforms at posv[j] + 2n ..., nu from the product column j - j0.  f' goes into that column (f itself is only written by
// the commit); used[j] = uniforms consumed behind the n normals, kv[j] = rejections, ierr[j] = what went wrong (0: nothing).
__global__ __launch_bounds__(256) void rs_verify_kernel(RsVerifyArgs a)
{
    __shared__ double red[4];
    const int64_t j = a.j0 + blockIdx.x;
    if ((uint64_t)j >= a.anchorP[0]) return;                   // not predicted
    const int64_t n = a.n;
    const double* fj = a.f + j * n;
    double* nj = a.nu + (j - a.j0) * n;
    const double* yj = a.y + j * n;
    const double* mj = a.mu + j * n;
    const uint64_t p0 = a.posv[j] + 2ull * (uint64_t)n;        // behind the n normals
    uint32_t uidx = 0;
    bool overflow = false, nan_state = false;
    auto next_u = [&]() -> double {
        const uint64_t q = p0 + uidx;
        double u = 0.5;
        if (q >= a.cap) overflow = true; else u = a.U[q];
        ++uidx;
        return u;
    };
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double yy = yj[i];
        if (yy != yy) continue;
        acc += ll_term(yy * (fj[i] + mj[i]));
    }
    const double ll0 = -block_sum_256(acc, red);
    const double u = next_u();
    const double log_y = ll0 + log(u);                                     // draw-f.cpp:28-29
    double eps_min = 0.0, eps_max = GP_2PI;                                // :33-34
    double eps = eps_min + (eps_max - eps_min) * next_u();                 // :35
    eps_min = eps - GP_2PI;                                                // :36
    int k = 0;
    double c, s;
    for (;;) {
        c = cos(eps);
        s = sin(eps);
        acc = 0.0;
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const double yy = yj[i];
            if (yy != yy) continue;
            const double fp = fj[i] * c + nj[i] * s;                       // :43
            acc += ll_term(yy * (fp + mj[i]));
        }
        const double llp = -block_sum_256(acc, red);
        if (llp > log_y) break;                                            // :45-47
        if (llp != llp) { nan_state = true; break; }                       // NaN state: never accepts
        if (eps < 0.0) eps_min = eps; else eps_max = eps;                  // :50-55
        if (eps_min == eps_max) eps = eps_min;                             // R::runif(a,a) = a
        else eps = eps_min + (eps_max - eps_min) * next_u();               // :56
        ++k;
        if (k >= 100000 || overflow) { overflow = true; break; }
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += 256) nj[i] = fj[i] * c + nj[i] * s;
    if (threadIdx.x == 0) {
        a.kv[j] = k;
        a.used[j] = uidx;
        a.ierr[j] = nan_state ? (int)GPIRT_E_NUMERIC : overflow ? (int)GPIRT_E_RNG : 0;
    }
}

// The same with the item's columns in registers (n <= NTH * EPT; rng_ess.hip's ess_kernel_reg does the item-keyed contract
// that way): f, nu, mu, y are read once, the (2 + k) likelihood passes run on registers.  Same per-element expression as
// rs_verify_kernel; the sums are block trees of NTH lanes instead of 256.  (No lambda touches the arrays: one that captures
// them by reference sends them to scratch memory -- tests/test_scratch_census.py.)
template <int EPT, int NTH>
__global__ __launch_bounds__(NTH) void rs_verify_reg_kernel(RsVerifyArgs a)
{
    __shared__ double red[8];
    const int64_t j = a.j0 + blockIdx.x;
    if ((uint64_t)j >= a.anchorP[0]) return;                   // not predicted
    const int64_t n = a.n;
    const double* fj = a.f + j * n;
    double* nj = a.nu + (j - a.j0) * n;
    const double* yj = a.y + j * n;
    const double* mj = a.mu + j * n;
    const uint64_t p0 = a.posv[j] + 2ull * (uint64_t)n;        // behind the n normals
    double F[EPT], V[EPT], M[EPT], Y[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + NTH * e;
        const bool in = i < n;
        F[e] = in ? fj[i] : 0.0; V[e] = in ? nj[i] : 0.0; M[e] = in ? mj[i] : 0.0;
        Y[e] = in ? yj[i] : __builtin_nan("");                  // NaN = skipped, like a missing response
    }
    uint32_t uidx = 0;
    bool overflow = false, nan_state = false;
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (Y[e] == Y[e]) acc += ll_term(Y[e] * (F[e] + M[e]));
    const double ll0 = -(NTH == 256 ? block_sum_256(acc, red) : block_sum_512(acc, red));
    double u = 0.5, u2 = 0.5;
    { const uint64_t q = p0 + uidx; if (q >= a.cap) overflow = true; else u = a.U[q]; ++uidx; }
    { const uint64_t q = p0 + uidx; if (q >= a.cap) overflow = true; else u2 = a.U[q]; ++uidx; }
    const double log_y = ll0 + log(u);                                     // draw-f.cpp:28-29
    double eps_min = 0.0, eps_max = GP_2PI;                                // :33-34
    double eps = eps_min + (eps_max - eps_min) * u2;                       // :35
    eps_min = eps - GP_2PI;                                                // :36
    int k = 0;
    double c, s;
    for (;;) {
        c = cos(eps);
        s = sin(eps);
        acc = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (Y[e] == Y[e]) acc += ll_term(Y[e] * ((F[e] * c + V[e] * s) + M[e]));   // :43
        const double llp = -(NTH == 256 ? block_sum_256(acc, red) : block_sum_512(acc, red));
        if (llp > log_y) break;                                            // :45-47
        if (llp != llp) { nan_state = true; break; }                       // NaN state: never accepts
        if (eps < 0.0) eps_min = eps; else eps_max = eps;                  // :50-55
        if (eps_min == eps_max) eps = eps_min;                             // R::runif(a,a) = a
        else {
            double un = 0.5;
            const uint64_t q = p0 + uidx; if (q >= a.cap) overflow = true; else un = a.U[q]; ++uidx;
            eps = eps_min + (eps_max - eps_min) * un;                      // :56
        }
        ++k;
        if (k >= 100000 || overflow) { overflow = true; break; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + NTH * e;
        if (i < n) nj[i] = F[e] * c + V[e] * s;
    }
    if (threadIdx.x == 0) {
        a.kv[j] = k;
        a.used[j] = uidx;
        a.ierr[j] = nan_state ? (int)GPIRT_E_NUMERIC : overflow ? (int)GPIRT_E_RNG : 0;
    }
}

// Walk the verified items in order: item j is exact when every item before it consumed what the predictor said.  Accept up
// to and including the first item whose own consumption differs (its start was exact), correct the next start, and leave
// the new exact state: anchor (item, start), the cursor, ctl[0] = first item NOT committed.  An error of an exact item is
// the draw's error (and closes the anchor, like the one-phase replay); errors behind a misprediction mean nothing.
__global__ __launch_bounds__(256) void rs_commit_scan_kernel(RsVerifyArgs a, uint64_t* anchor, uint64_t* pos, uint64_t* ctl, int* err)
{
    __shared__ unsigned long long first;
    const int64_t j0 = a.j0;
    const int64_t jp = (int64_t)a.anchorP[0];
    if (threadIdx.x == 0) first = (unsigned long long)jp;
    __syncthreads();
    const uint64_t step = 2ull * (uint64_t)a.n;
    for (int64_t j = j0 + threadIdx.x; j < jp; j += 256) {
        const bool off = a.posv[j] + step + (uint64_t)a.used[j] != a.posv[j + 1];
        if (off || a.ierr[j] != 0) atomicMin(&first, (unsigned long long)j);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int64_t ndone = jp;
    if ((int64_t)first < jp) {
        const int64_t js = (int64_t)first;
        if (a.ierr[js] != 0) {                                  // a genuine failure of an exact item
            atomicCAS(err, 0, a.ierr[js]);
            ctl[0] = (uint64_t)js;
            anchor[0] = (uint64_t)a.m;
            return;
        }
        ndone = js + 1;
        a.posv[ndone] = a.posv[js] + step + (uint64_t)a.used[js];
        ctl[1] += 1;                                            // mispredictions so far (gpirt_sampler_get "rs_stats")
    }
    ctl[0] = (uint64_t)ndone;
    anchor[0] = (uint64_t)ndone;
    anchor[1] = a.posv[ndone];
    *pos = a.posv[ndone];
}

// f[:, j] = the verified draw, ess_k[j] = its rejection count, for the committed items j0 <= j < ctl[0]
__global__ __launch_bounds__(256) void rs_commit_copy_kernel(RsVerifyArgs a, const uint64_t* __restrict__ ctl, double* __restrict__ f, int* __restrict__ k_out)
{
    const int64_t j = a.j0 + blockIdx.x;
    if ((uint64_t)j >= ctl[0]) return;
    const double* src = a.nu + (j - a.j0) * a.n;
    double* dst = f + j * a.n;
    for (int64_t i = threadIdx.x; i < a.n; i += 256) dst[i] = src[i];
    if (threadIdx.x == 0) k_out[j] = a.kv[j];
}

}  // namespace

int launch_rs32_tiles(hipStream_t stream, const double* L, int64_t n, int64_t ldl, float* Lt, bool diag_only)
{
    const int64_t nrg = (n + RS_ROWS - 1) / RS_ROWS, nk8 = rs32_tile_octs(n);
    hipLaunchKernelGGL(rs32_tile_kernel, dim3(diag_only ? 2 : 8, (unsigned)nrg), dim3(256), 0, stream, L, n, ldl, nk8, Lt, diag_only ? 1 : 0);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs3p_products(hipStream_t stream, const Rs3Args& a)
{
    if (a.nunits <= 0) return 0;
    if (a.lr) hipLaunchKernelGGL(rs3p_products_lr_kernel<8>, dim3((unsigned)a.nunits), dim3(512), 0, stream, a);
    else      hipLaunchKernelGGL(rs3p_products_kernel, dim3((unsigned)a.nunits), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs3p_decide(hipStream_t stream, const Rs3Args& a)
{
    static bool attr_set = false;
    constexpr int pad = 96 * 1024;        // with the static LDS: more than half a compute unit's 160 KB -> one work-group per CU
    if (!attr_set) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rs3p_decide_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, pad));
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rs3p_decide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, pad));
        attr_set = true;
    }
    if (a.lr) hipLaunchKernelGGL(rs3p_decide_kernel<true>, dim3(RS3_CAND * PD_PARTS), dim3(320), pad, stream, a);
    else      hipLaunchKernelGGL(rs3p_decide_kernel<false>, dim3(RS3_CAND * PD_PARTS), dim3(320), pad, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

// the predictor's work-groups: every (row group, part of RS3P_KC columns) that holds a non-zero of L, full parts first
void rs3p_unit_table(int64_t n, std::vector<uint32_t>& units, int* nfull)
{
    std::vector<uint32_t> ragged;
    units.clear();
    const int64_t nbx = (n + RS_ROWS - 1) / RS_ROWS;
    for (int64_t bx = 0; bx < nbx; ++bx) {
        const int64_t kall = ((bx + 1) * RS_ROWS < n) ? (bx + 1) * RS_ROWS : n;
        for (int64_t by = 0; by * RS3P_KC < kall; ++by) {
            const uint32_t u = (uint32_t)bx | ((uint32_t)by << 16);
            if ((by + 1) * RS3P_KC <= kall) units.push_back(u); else ragged.push_back(u);
        }
    }
    *nfull = (int)units.size();
    units.insert(units.end(), ragged.begin(), ragged.end());
}

int launch_rs_pred_start(hipStream_t stream, const uint64_t* anchor, uint64_t* anchorP, const int* k_last, int64_t m)
{
    hipLaunchKernelGGL(rs_pred_start_kernel, dim3(1), dim3(256), 0, stream, anchor, anchorP, k_last, m);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs_gather(hipStream_t stream, const double* Nrm, const uint64_t* posv, const uint64_t* anchorP, int64_t n, int64_t j0,
                     int64_t m, double* Z)
{
    if (m <= j0) return 0;
    hipLaunchKernelGGL(rs_gather_kernel, dim3((unsigned)(m - j0)), dim3(256), 0, stream, Nrm, posv, anchorP, n, j0, m, Z);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs_verify(hipStream_t stream, const RsVerifyArgs& a)
{
    if (a.m <= a.j0) return 0;
    if (a.n <= 256 * 8)       hipLaunchKernelGGL((rs_verify_reg_kernel<8, 256>), dim3((unsigned)(a.m - a.j0)), dim3(256), 0, stream, a);
    else if (a.n <= 512 * 16) hipLaunchKernelGGL((rs_verify_reg_kernel<16, 512>), dim3((unsigned)(a.m - a.j0)), dim3(512), 0, stream, a);
    else                      hipLaunchKernelGGL(rs_verify_kernel, dim3((unsigned)(a.m - a.j0)), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs_commit(hipStream_t stream, const RsVerifyArgs& a, uint64_t* anchor, uint64_t* pos, uint64_t* ctl, int* err, double* f,
                     int* k_out)
{
    if (a.m <= a.j0) return 0;
    hipLaunchKernelGGL(rs_commit_scan_kernel, dim3(1), dim3(256), 0, stream, a, anchor, pos, ctl, err);
    hipLaunchKernelGGL(rs_commit_copy_kernel, dim3((unsigned)(a.m - a.j0)), dim3(256), 0, stream, a, ctl, f, k_out);
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
