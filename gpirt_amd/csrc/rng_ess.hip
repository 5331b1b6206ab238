// rng_ess.hip -- random-number fills, ll_bar() and the elliptical slice sampler.
//
//   item_fill        GPIRT_RNG_ITEM uniforms / normals (Philox4x32-10 sub-streams)
//   rstream_normals  R::rnorm(0,1) replay (src/mvnormal.h:7-9) from a pre-generated uniform stream
//   ll_bar_kernel    src/log-likelihood.cpp:25-37 for every column
//   ess_kernel       ess(), src/draw-f.cpp:21-60: ONE work-group per item column; the (2+k) passes
//                    of ll_bar are block reductions with a fixed tree, every lane carries the same
//                    bracket state, so control flow is work-group uniform.  HBM/L2 bound.
#include "common.h"
#include "kernels.h"
#include "ll_fast.h"

namespace gpirt {

namespace {

// one term of ll(): as written (library exp and log) or the 70-instruction form of ll_fast.h (FAST; the slice kernels of
// the item-keyed RNG -- an R-stream replay keeps the reference's formula to the letter)
template <bool FAST>
__device__ __forceinline__ double ll_t(double a) { return FAST ? ll_term_fast(a) : ll_term(a); }

__global__ void ll_term_probe_kernel(const double* __restrict__ a, int64_t n, double* __restrict__ out, int fast)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
        out[g] = fast ? ll_term_fast(a[g]) : ll_term(a[g]);
}

__global__ void item_fill_kernel(uint64_t seed, uint32_t iter, uint32_t stage, uint32_t item0,
                                 int64_t n_items, int64_t n_index, double* __restrict__ out,
                                 bool normal)
{
    const int64_t total = n_items * n_index;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t it = g / n_index, ix = g - it * n_index;
        const double u = item_uniform(seed, iter, stage, item0 + (uint32_t)it, (uint32_t)ix);
        out[g] = normal ? qnorm_as241(u) : u;
    }
}

__global__ void rstream_normals_kernel(const double* __restrict__ U, const uint64_t* __restrict__ pos,
                                       int64_t col_stride, int64_t n, int64_t m,
                                       double* __restrict__ out)
{
    const uint64_t p = *pos;
    const int64_t total = n * m;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = g / n, i = g - j * n;
        const uint64_t q = p + (uint64_t)(j * col_stride + 2 * i);
        out[g] = rnorm_from_two(U[q], U[q + 1]);
    }
}

__global__ __launch_bounds__(256) void ll_bar_kernel(const double* __restrict__ f,
                                                     const double* __restrict__ y,
                                                     const double* __restrict__ mu, int64_t n,
                                                     double* __restrict__ out)
{
    __shared__ double red[4];
    const int64_t j = blockIdx.x;
    const double* fj = f + j * n;
    const double* yj = y + j * n;
    const double* mj = mu ? mu + j * n : nullptr;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double yy = yj[i];
        if (yy != yy) continue;
        const double g = mj ? fj[i] + mj[i] : fj[i];
        acc += ll_term(yy * g);
    }
    const double s = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[j] = -s;
}

constexpr int ESS_MAX_TRIALS = 100000;

template <bool FAST>
__global__ __launch_bounds__(256) void ess_kernel(EssArgs a)
{
    __shared__ double red[4];
    const int64_t j = blockIdx.x;
    const int64_t n = a.n;
    double* fj = a.f + j * n;
    const double* nj = a.nu + j * n;
    const double* yj = a.y + j * n;
    const double* mj = a.mu + j * n;
    const bool stream = (a.U != nullptr);
    const uint64_t p0 = stream ? (*a.pos + 2ull * (uint64_t)n) : 0ull;   // after the n normals
    const uint32_t item = a.item0 + (uint32_t)j;
    uint32_t uidx = 0;
    bool overflow = false, nan_state = false;
    auto next_u = [&]() -> double {
        double u;
        if (stream) {
            const uint64_t q = p0 + uidx;
            if (q >= a.cap) { overflow = true; u = 0.5; } else u = a.U[q];
        } else {
            u = item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx);
        }
        ++uidx;
        return u;
    };

    // log_y = ll_bar(f, y, mu) + log(u)                                   draw-f.cpp:28-29
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double yy = yj[i];
        if (yy != yy) continue;
        acc += ll_t<FAST>(yy * (fj[i] + mj[i]));
    }
    const double ll0 = -block_sum_256(acc, red);
    const double u = next_u();
    const double log_y = ll0 + log(u);
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
            acc += ll_t<FAST>(yy * (fp + mj[i]));
        }
        const double llp = -block_sum_256(acc, red);
        if (llp > log_y) break;                                            // :45-47
        if (llp != llp) { nan_state = true; break; }                       // NaN state: never accepts
        if (eps < 0.0) eps_min = eps; else eps_max = eps;                  // :50-55
        if (eps_min == eps_max) eps = eps_min;                             // R::runif(a,a) = a
        else eps = eps_min + (eps_max - eps_min) * next_u();               // :56
        ++k;
        if (k >= ESS_MAX_TRIALS || overflow) { overflow = true; break; }
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += 256) fj[i] = fj[i] * c + nj[i] * s;
    if (threadIdx.x == 0) {
        if (a.k_out) a.k_out[j] = k;
        if (nan_state && a.err) atomicCAS(a.err, 0, (int)GPIRT_E_NUMERIC);
        else if (overflow && a.err) atomicCAS(a.err, 0, stream ? GPIRT_E_RNG : GPIRT_E_NUMERIC);
        if (stream) *a.pos = p0 + uidx;
    }
}

// Register-resident variant for n <= NTH * EPT: each lane keeps its EPT entries of f, nu, mu and y in
// registers, so the (2 + k) likelihood passes of a column touch memory once; arithmetic is identical
// to ess_kernel (same per-element expression, same reduction tree).
template <int EPT, int NTH, bool FAST>
__global__ __launch_bounds__(NTH) void ess_kernel_reg(EssArgs a)
{
    __shared__ double red[8];
    auto block_sum = [&](double v) { return NTH == 256 ? block_sum_256(v, red) : block_sum_512(v, red); };
    const int64_t j = blockIdx.x;
    const int64_t n = a.n;
    double* fj = a.f + j * n;
    const double* nj = a.nu + j * n;
    const double* yj = a.y + j * n;
    const double* mj = a.mu + j * n;
    const uint32_t item = a.item0 + (uint32_t)j;
    double F[EPT], V[EPT], M[EPT], Y[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + NTH * e;
        const bool in = i < n;
        F[e] = in ? fj[i] : 0.0;
        V[e] = in ? nj[i] : 0.0;
        M[e] = in ? mj[i] : 0.0;
        Y[e] = in ? yj[i] : __builtin_nan("");      // NaN = skipped, like a missing response
    }
    uint32_t uidx = 0;
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (Y[e] == Y[e]) acc += ll_t<FAST>(Y[e] * (F[e] + M[e]));
    const double ll0 = -block_sum(acc);
    const double u = item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
    const double log_y = ll0 + log(u);                                     // draw-f.cpp:28-29
    double eps_min = 0.0, eps_max = GP_2PI;
    double eps = eps_min + (eps_max - eps_min) * item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
    eps_min = eps - GP_2PI;                                                // :36
    int k = 0;
    bool bad = false;
    double c, s;
    for (;;) {
        c = cos(eps);
        s = sin(eps);
        acc = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (Y[e] == Y[e]) acc += ll_t<FAST>(Y[e] * ((F[e] * c + V[e] * s) + M[e]));   // :43
        const double llp = -block_sum(acc);
        if (llp > log_y) break;                                            // :45-47
        if (llp != llp) { bad = true; break; }
        if (eps < 0.0) eps_min = eps; else eps_max = eps;                  // :50-55
        if (eps_min == eps_max) eps = eps_min;
        else eps = eps_min + (eps_max - eps_min) * item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
        ++k;
        if (k >= ESS_MAX_TRIALS) { bad = true; break; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + NTH * e;
        if (i < n) fj[i] = F[e] * c + V[e] * s;
    }
    if (threadIdx.x == 0) {
        if (a.k_out) a.k_out[j] = k;
        if (bad && a.err) atomicCAS(a.err, 0, (int)GPIRT_E_NUMERIC);
    }
}

__global__ void advance_pos_kernel(uint64_t* pos, uint64_t delta) { *pos += delta; }

}  // namespace

int launch_item_uniforms(hipStream_t stream, uint64_t seed, uint32_t iter, uint32_t stage,
                         uint32_t item0, int64_t n_items, int64_t n_index, double* out, bool normal)
{
    const int64_t total = n_items * n_index;
    if (total <= 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(item_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, iter,
                       stage, item0, n_items, n_index, out, normal);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rstream_normals(hipStream_t stream, const double* U, const uint64_t* d_pos,
                           int64_t col_stride, int64_t n, int64_t m, double* out)
{
    const int64_t total = n * m;
    if (total <= 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rstream_normals_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, U, d_pos,
                       col_stride, n, m, out);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_ll_bar(hipStream_t stream, const double* f, const double* y, const double* mu, int64_t n,
                  int64_t m, double* out)
{
    if (m <= 0) return 0;
    hipLaunchKernelGGL(ll_bar_kernel, dim3((unsigned)m), dim3(256), 0, stream, f, y, mu, n, out);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_ess(hipStream_t stream, const EssArgs& a)
{
    if (a.m <= 0) return 0;
    // a.ll_exact (GPIRT_LL_EXACT=1): log(1 + exp(-a)) through the library in every mode
    const bool fast = a.U == nullptr && a.ll_exact != 1;
    if (a.U == nullptr && a.n <= 256 * 8) {
        if (fast) hipLaunchKernelGGL((ess_kernel_reg<8, 256, true>), dim3((unsigned)a.m), dim3(256), 0, stream, a);
        else      hipLaunchKernelGGL((ess_kernel_reg<8, 256, false>), dim3((unsigned)a.m), dim3(256), 0, stream, a);
    } else if (a.U == nullptr && a.n <= 256 * 32) {
        if (fast) hipLaunchKernelGGL((ess_kernel_reg<16, 512, true>), dim3((unsigned)a.m), dim3(512), 0, stream, a);
        else      hipLaunchKernelGGL((ess_kernel_reg<16, 512, false>), dim3((unsigned)a.m), dim3(512), 0, stream, a);
    } else {
        if (fast) hipLaunchKernelGGL(ess_kernel<true>, dim3((unsigned)a.m), dim3(256), 0, stream, a);
        else      hipLaunchKernelGGL(ess_kernel<false>, dim3((unsigned)a.m), dim3(256), 0, stream, a);
    }
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_ll_term_probe(hipStream_t stream, const double* a, int64_t n, double* out, bool fast)
{
    if (n <= 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ll_term_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, n, out, fast ? 1 : 0);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_advance_pos(hipStream_t stream, uint64_t* pos, uint64_t delta)
{
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, stream, pos, delta);
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
