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

namespace gpirt {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

// L -> single-precision tiles in the order the predictor's products read them: tile (row group rg of 32 rows, oct ko of 8
// columns) = 1 KiB, lane l = (row rg 32 + (l & 31), columns 8 ko + 4 (l >> 5) .. + 3) as one float4: a wave's step is one
// contiguous kilobyte, its steps follow each other in memory.  Rows and columns past the matrix are zeros; the strict upper
// triangle of L holds zeros (gpirt_sampler_create).  Only the tiles a product reads are written.
__global__ __launch_bounds__(256) void rs32_tile_kernel(const double* __restrict__ L, int64_t n, int64_t ldl, int64_t nk8, float* __restrict__ Lt)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rg = blockIdx.y;
    const int64_t r0 = rg * RS_ROWS;
    const int64_t kall = (r0 + RS_ROWS < n) ? r0 + RS_ROWS : n;
    const int64_t ko_end = (kall + 7) / 8;
    const int64_t row = r0 + (lane & 31);
    for (int64_t ko = (int64_t)blockIdx.x * 4 + wave; ko < ko_end; ko += (int64_t)gridDim.x * 4) {
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
constexpr int P32_STEPS = RS_KC / 32;
template <bool FULL>
__device__ __forceinline__ void rs3p_product_unit(const Rs3Args& a, const uint64_t base, const int bx, const int by, float* lds)
{
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6, c = lane & 31, hh = lane >> 5;
    const int64_t n = a.n;
    const int64_t r0 = (int64_t)bx * RS_ROWS;
    const int64_t kall = (r0 + RS_ROWS < n) ? r0 + RS_ROWS : n;
    const int64_t k0 = (int64_t)by * RS_KC;
    if (!FULL && k0 >= kall) return;
    const int64_t kend = FULL ? k0 + RS_KC : ((k0 + RS_KC < kall) ? k0 + RS_KC : kall);          // this part's columns [k0, kend)
    const int kw = (int)(kend - k0);
    // this wave's steps: octs of 8 columns, a quarter of RS_KC per wave
    const int64_t o_beg = (k0 >> 3) + (int64_t)kq * P32_STEPS;
    int steps = P32_STEPS;
    if (!FULL) {
        int64_t o_end = o_beg + P32_STEPS;
        const int64_t o_all = (kend + 7) >> 3;
        if (o_end > o_all) o_end = o_all;
        steps = o_end > o_beg ? (int)(o_end - o_beg) : 0;
    }
    const float* Lp = a.Lt32 + ((int64_t)bx * a.nk8 + o_beg) * 256 + 4 * lane;
    float* W0 = lds; float* W1 = lds + RS_KC; float* W2 = W1 + 2 * RS_KC + 16;
    static_assert(RS_KC + (2 * RS_KC + 16) + (2 * RS_KC + 32) <= 4 * 1024 + 64, "the windows must fit");
    const uint64_t item_step = 2ull * (uint64_t)n + 2ull;
    const double* N0 = a.Nrm + base + 2ull * (uint64_t)k0;
    const double* N1 = N0 + item_step;
    const double* N2 = N1 + item_step;
    float4 lv[P32_STEPS];
    if (FULL) {
        // the windows' loads go out first (L2 hits), the wave's sixteen kilobytes of L straight behind them
        constexpr int C0 = RS_KC / 256, C1 = (2 * RS_KC + 16 + 255) / 256, C2 = (2 * RS_KC + 32 + 255) / 256;
        double w0[C0], w1[C1], w2[C2];
#pragma unroll
        for (int q = 0; q < C0; ++q) w0[q] = N0[2 * (tid + 256 * q)];
#pragma unroll
        for (int q = 0; q < C1; ++q) { const int x = tid + 256 * q; w1[q] = N1[x < 2 * RS_KC + 16 ? x : 0]; }
#pragma unroll
        for (int q = 0; q < C2; ++q) { const int x = tid + 256 * q; w2[q] = N2[x < 2 * RS_KC + 32 ? x : 0]; }
#pragma unroll
        for (int u = 0; u < P32_STEPS; ++u) lv[u] = *reinterpret_cast<const float4*>(Lp + (int64_t)u * 256);
#pragma unroll
        for (int q = 0; q < C0; ++q) W0[tid + 256 * q] = (float)w0[q];
#pragma unroll
        for (int q = 0; q < C1; ++q) { const int x = tid + 256 * q; if (x < 2 * RS_KC + 16) W1[x] = (float)w1[q]; }
#pragma unroll
        for (int q = 0; q < C2; ++q) { const int x = tid + 256 * q; if (x < 2 * RS_KC + 32) W2[x] = (float)w2[q]; }
    } else {
        // (entries of the windows this part can touch; a ragged last oct reads up to 7 columns past the part -- zeros in the
        // tile, so the normals there must be staged too: an unstaged LDS word may hold a NaN)
        const int kw8 = (kw + 7) & ~7;
        const int x1 = 2 * kw8 + 16, x2 = 2 * kw8 + 32;
        for (int x = tid; x < kw8; x += 256) W0[x] = (float)N0[2 * x];
        for (int x = tid; x < x1; x += 256) W1[x] = (float)N1[x];
        for (int x = tid; x < x2; x += 256) W2[x] = (float)N2[x];
#pragma unroll
        for (int u = 0; u < P32_STEPS; ++u) lv[u] = (u < steps) ? *reinterpret_cast<const float4*>(Lp + (int64_t)u * 256) : float4{ 0.f, 0.f, 0.f, 0.f };
    }
    __syncthreads();
    // lane's candidate: column kk of the part is at Zb[zs * kk]
    const float* Zb = (c == 0) ? W0 : (c < 16 ? W1 + (c - 1) : W2 + (c - 16));
    const int zs = (c == 0) ? 1 : 2;
    const float* z = Zb + zs * ((int)((o_beg << 3) - k0) + 4 * hh);
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < P32_STEPS; ++s) {
        if (FULL || s < steps) {                                 // (uniform over the wave)
            const float z0 = z[zs * (8 * s)], z1 = z[zs * (8 * s + 1)], z2 = z[zs * (8 * s + 2)], z3 = z[zs * (8 * s + 3)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z0, lv[s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z1, lv[s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z2, lv[s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z3, lv[s].w, acc, 0, 0, 0);
        }
    }
    // acc[4 q + r] = candidate 8 q + 4 hh + r, row r0 + c.  The quarters meet in LDS: red[kq][cand][row]
    __syncthreads();                                            // (the windows are no longer read)
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
}

__global__ __launch_bounds__(256, 4) void rs3p_products_kernel(Rs3Args a)
{
    // windows: W0 RS_KC floats | W1 2 RS_KC + 16 | W2 2 RS_KC + 32; then the four waves' 32 x 32 accumulators (16 KB)
    __shared__ __attribute__((aligned(16))) float lds[4 * 1024 + 64];
    const uint64_t item0 = a.anchor[0], base = a.anchor[1], stalled = a.anchor[3];
    const uint32_t unit = a.units[blockIdx.x];
    if (item0 >= (uint64_t)a.m || stalled != 0) return;       // every item is predicted, or the predictor has stalled
    const int bx = (int)(unit & 0xffffu), by = (int)(unit >> 16);
    if ((int)blockIdx.x < a.nfull) rs3p_product_unit<true>(a, base, bx, by, lds);
    else                           rs3p_product_unit<false>(a, base, bx, by, lds);
}

// the predictor starts where the exact state stands
__global__ void rs_pred_start_kernel(const uint64_t* __restrict__ anchor, uint64_t* __restrict__ anchorP)
{
    anchorP[0] = anchor[0]; anchorP[1] = anchor[1]; anchorP[2] = anchor[2]; anchorP[3] = 0;
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

int launch_rs32_tiles(hipStream_t stream, const double* L, int64_t n, int64_t ldl, float* Lt)
{
    const int64_t nrg = (n + RS_ROWS - 1) / RS_ROWS, nk8 = rs32_tile_octs(n);
    hipLaunchKernelGGL(rs32_tile_kernel, dim3(8, (unsigned)nrg), dim3(256), 0, stream, L, n, ldl, nk8, Lt);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs3p_products(hipStream_t stream, const Rs3Args& a)
{
    if (a.nunits <= 0) return 0;
    hipLaunchKernelGGL(rs3p_products_kernel, dim3((unsigned)a.nunits), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs_pred_start(hipStream_t stream, const uint64_t* anchor, uint64_t* anchorP)
{
    hipLaunchKernelGGL(rs_pred_start_kernel, dim3(1), dim3(1), 0, stream, anchor, anchorP);
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
    hipLaunchKernelGGL(rs_verify_kernel, dim3((unsigned)(a.m - a.j0)), dim3(256), 0, stream, a);
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
