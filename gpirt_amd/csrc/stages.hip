// stages.hip -- the element-wise / reduction kernels around the GEMM core:
//   colnorm_s        s = 1 - sqrt(colsum(tmp % tmp))                 src/draw-fstar.cpp:20  (quirk Q2)
//   fstar_epilogue   f*_ij = R::rnorm(mean_ij + mu*_ij, s_i)         src/draw-fstar.cpp:25-28
//   indicators / loglik_terms / theta_sample                         src/draw-theta.cpp:3-37 as a GEMM
//   draw_beta        random-walk MH per item, two coefficients       src/draw-beta.cpp:3-41
//   linear_mean      mu = X beta, X = [1, x]                         src/gpirtMCMC.cpp:33,40,74-75,93-94
// All are HBM/L2-bound streaming kernels: one work-group per column (item, grid point or
// respondent), coalesced along the contiguous column, fixed-tree block reductions.
#include "common.h"
#include "kernels.h"

namespace gpirt {

namespace {

#define GP_LN_SQRT_2PI 0.918938533204672741780329736406

// R::rnorm(mu, sd) given a standard normal z (z is only consumed in the last branch)
__device__ __forceinline__ double r_rnorm(double mu, double sd, double z, bool& consumes)
{
    consumes = false;
    if (mu != mu || !isfinite(sd) || sd < 0.0) return __builtin_nan("");
    if (sd == 0.0 || !isfinite(mu)) return mu;
    consumes = true;
    return mu + sd * z;
}

__device__ __forceinline__ double r_dnorm_log(double x, double mu, double sd)
{
    const double z = fabs((x - mu) / sd);
    return -(GP_LN_SQRT_2PI + 0.5 * z * z + log(sd));
}

// ------------------------------------------------------------------ draw_fstar -------------
__global__ __launch_bounds__(256) void colnorm_s_kernel(const double* __restrict__ tmp, int64_t n,
                                                        int64_t ld, double* __restrict__ s)
{
    __shared__ double red[4];
    const double* c = tmp + (int64_t)blockIdx.x * ld;
    double acc = 0.0;
    for (int64_t k = threadIdx.x; k < n; k += 256) { const double v = c[k]; acc += v * v; }
    const double q = block_sum_256(acc, red);
    if (threadIdx.x == 0) s[blockIdx.x] = 1.0 - sqrt(q);
}

// s_j = 1 - sqrt(v_j^T G v_j), G = B^T B (r x r, ld ldg): draw-fstar.cpp:20 through the rank-r form of K*
__global__ __launch_bounds__(128) void lowrank_s_kernel(const double* __restrict__ V, int64_t N, int r,
                                                        const double* __restrict__ G, int64_t ldg, double* __restrict__ s)
{
    __shared__ double sv[128], red[128];
    const int64_t j = blockIdx.x;
    const int k = threadIdx.x;
    sv[k] = (k < r) ? V[j + (int64_t)k * N] : 0.0;
    __syncthreads();
    double t = 0.0;
    if (k < r) {
        for (int l = 0; l < r; ++l) t += G[k + (int64_t)l * ldg] * sv[l];
        t *= sv[k];
    }
    red[k] = t;
    __syncthreads();
    for (int w = 64; w > 0; w >>= 1) {
        if (k < w) red[k] += red[k + w];
        __syncthreads();
    }
    // the reference's sum of squares cannot be negative; the quadratic form can, by cancellation, when it is ~0
    if (k == 0) s[j] = 1.0 - sqrt(fmax(red[0], 0.0));
}

// exclusive count of RNG-consuming grid points (s_i > 0 and finite) -- R-stream replay only.  One work-group: every thread
// counts a contiguous chunk, the chunks' counts are scanned in LDS, every thread writes its chunk's offsets.
__global__ __launch_bounds__(256) void fstar_offsets_kernel(const double* __restrict__ s, int N, int* __restrict__ off)
{
    __shared__ int cnt[256];
    const int tid = threadIdx.x;
    const int per = (N + 255) / 256;
    const int i0 = tid * per, i1 = (i0 + per < N) ? i0 + per : N;
    int c = 0;
    for (int i = i0; i < i1; ++i) { const double v = s[i]; c += (isfinite(v) && v > 0.0) ? 1 : 0; }
    cnt[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int q = 0; q < 256; ++q) { const int t = cnt[q]; cnt[q] = run; run += t; }
        off[N] = run;
    }
    __syncthreads();
    c = cnt[tid];
    for (int i = i0; i < i1; ++i) {
        off[i] = c;
        const double v = s[i];
        if (isfinite(v) && v > 0.0) ++c;
    }
}

__global__ __launch_bounds__(256) void fstar_epilogue_kernel(FstarEpiArgs a, const int* __restrict__ off)
{
    const int64_t j = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    const double mean = a.mean[i + j * a.N] + a.mu_star[i + j * a.N];       // :25
    const double sd = a.s[i];
    double z;
    if (a.U) {
        const uint64_t q = *a.pos + 2ull * ((uint64_t)j * (uint64_t)off[a.N] + (uint64_t)off[i]);
        if (q + 1 >= a.cap) { if (a.err) atomicCAS(a.err, 0, (int)GPIRT_E_RNG); z = 0.0; }
        else z = rnorm_from_two(a.U[q], a.U[q + 1]);
    } else {
        z = qnorm_as241(item_uniform(a.seed, a.iter, GPIRT_ST_FSTAR, a.item0 + (uint32_t)j, (uint32_t)i));
    }
    bool consumes;
    a.out[i + j * a.N] = r_rnorm(mean, sd, z, consumes);                    // :27
    if (a.mean_out) a.mean_out[i + j * a.N] = mean;
}

__global__ void fstar_advance_kernel(uint64_t* pos, const int* off, int N, int64_t m)
{
    *pos += 2ull * (uint64_t)off[N] * (uint64_t)m;
}

// ------------------------------------------------------------------ draw_theta -------------
__global__ void indicators_kernel(const double* __restrict__ y, int64_t total, double* __restrict__ Ypm)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const double v = y[g];
        Ypm[g] = (v == 1.0) ? 1.0 : 0.0;
        Ypm[g + total] = (v == -1.0) ? 1.0 : 0.0;
    }
}

// G+[k,j] = -log(1+exp(-f*_kj))  (y = +1),  G-[k,j] = -log(1+exp(+f*_kj))  (y = -1)
// Gpm has leading dimension ldg >= N (rows N .. ldg-1 are padding the GEMM may read but never uses)
__global__ void loglik_terms_kernel(const double* __restrict__ fstar, int64_t N, int64_t m, double* __restrict__ Gpm,
                                    int64_t ldg, const int* __restrict__ run_if)
{
    if (run_if != nullptr && *run_if == 0) return;
    const int64_t total = N * m;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const double v = fstar[g];
        const int64_t o = (g % N) + (g / N) * ldg;
        // |f*| > ~709.8 overflows exp(): the reference adds -inf for the responses that see it and skips the others
        // (isnan(y), src/log-likelihood.cpp:16); in the GEMM form the skipped ones would be 0 * -inf = NaN, so -inf
        // is held at a finite -1e300 (exp() of the sum is still exactly 0).  NaN stays NaN.
        double gp = -ll_term(1.0 * v), gm = -ll_term(-1.0 * v);
        if (gp == -INFINITY) gp = -1e300;
        if (gm == -INFINITY) gm = -1e300;
        Gpm[o] = gp;
        Gpm[o + ldg * m] = gm;
    }
}

// one work-group per respondent; 256 threads x 4 consecutive grid points cover N <= 1024
__global__ __launch_bounds__(256) void theta_sample_kernel(ThetaArgs a)
{
    __shared__ double red[8];
    __shared__ double scan[256];
    __shared__ int    ired[4];
    const int64_t ib = blockIdx.x;                 // column of logpost
    const int64_t i = ib + a.i0;                   // respondent
    const double* lp = a.logpost + ib * a.N;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int N = (int)a.N;
    double P[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = t * 4 + e;
        if (k < N) {
            const double ts = -5.0 + (double)k * 0.01;
            P[e] = r_dnorm_log(ts, 0.0, 1.0) + lp[k];                      // draw-theta.cpp:18
        } else P[e] = -INFINITY;
    }
    if (a.stabilise) {
        double mx = fmax(fmax(P[0], P[1]), fmax(P[2], P[3]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
        if (lane == 0) red[w] = mx;
        __syncthreads();
        mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) P[e] -= mx;
    }
    // exp, cumsum                                                          :21-22
    double run = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = t * 4 + e;
        const double v = (k < N) ? exp(P[e]) : 0.0;
        run += v;
        P[e] = run;
    }
    scan[t] = run;
    __syncthreads();
    // Hillis-Steele inclusive scan of the 256 thread totals
    for (int d = 1; d < 256; d <<= 1) {
        const double add = (t >= d) ? scan[t - d] : 0.0;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    const double base = (t > 0) ? scan[t - 1] : 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) P[e] += base;
    // max / min of the CDF                                                 :23-24
    double mx = -INFINITY, mn = INFINITY;
    bool has_nan = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = t * 4 + e;
        if (k < N) { mx = fmax(mx, P[e]); mn = fmin(mn, P[e]); has_nan |= (P[e] != P[e]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, off, 64));
        mn = fmin(mn, __shfl_xor(mn, off, 64));
    }
    if (lane == 0) { red[w] = mx; red[4 + w] = mn; }
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    mn = fmin(fmin(red[4], red[5]), fmin(red[6], red[7]));
    const double range = mx - mn;
    double u;
    if (a.U) u = a.U[*a.pos + (uint64_t)i];
    else u = item_uniform(a.seed, a.iter, GPIRT_ST_THETA, (uint32_t)i, 0);  // :27
    int first = 0x7fffffff;
#pragma unroll
    for (int e = 3; e >= 0; --e) {
        const int k = t * 4 + e;
        if (k < N) {
            const double cdf = (P[e] - mn) / range;                         // :25
            if (cdf > u) first = k;                                         // :29-34
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
    if (lane == 0) ired[w] = first;
    __syncthreads();
    if (t == 0) {
        first = min(min(ired[0], ired[1]), min(ired[2], ired[3]));
        if (first == 0x7fffffff) {
            a.theta_out[i] = __builtin_nan("");       // reference reads theta_star[N]: out of bounds
            if (a.degenerate) atomicAdd(a.degenerate, 1);
        } else {
            a.theta_out[i] = -5.0 + (double)first * 0.01;
        }
    }
    (void)has_nan;
}

// ------------------------------------------------------------------ draw_beta --------------
__global__ __launch_bounds__(256) void draw_beta_kernel(BetaArgs a)
{
    __shared__ double red[4];
    const int64_t j = blockIdx.x;
    const int64_t n = a.n;
    const double* fj = a.f + j * n;
    const double* yj = a.y + j * n;
    const uint32_t item = a.item0 + (uint32_t)j;
    double cv[2] = { a.beta[0 + 2 * j], a.beta[1 + 2 * j] };
    double pv[2] = { cv[0], cv[1] };
    uint64_t q = 0;
    if (a.U) q = *a.pos + a.item_off[j];
    double kept_ll = 0.0;                 // ll(y | current beta) after the last accept / reject
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double step = a.step[k + 2 * j];
        double z;
        if (a.U) z = rnorm_from_two(a.U[q], a.U[q + 1]);
        else z = qnorm_as241(item_uniform(a.seed, a.iter, GPIRT_ST_BETA, item, (uint32_t)(2 * k)));
        bool consumes;
        pv[k] = r_rnorm(cv[k], step, z, consumes);                          // draw-beta.cpp:22
        if (a.U && consumes) q += 2;
        const double pm = a.pm[k + 2 * j], ps = a.ps[k + 2 * j];
        const double pv_prior = r_dnorm_log(pv[k], pm, ps);                 // :25
        const double cv_prior = r_dnorm_log(cv[k], pm, ps);                 // :26
        // ll(current) of the second coefficient is a sum already formed in the first step: the proposal's if it was
        // accepted (same mu expression, same reduction tree), the current one's otherwise -- 3 passes over y, f instead of 4
        double accp = 0.0, accc = 0.0;
        const bool reuse = (k == 1);
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const double yy = yj[i];
            if (yy != yy) continue;
            const double th = a.theta[i], f = fj[i];
            const double mup = pv[0] + th * pv[1];
            accp += ll_term(yy * (f + mup));                                // :27
            if (!reuse) {
                const double muc = cv[0] + th * cv[1];
                accc += ll_term(yy * (f + muc));                            // :28
            }
        }
        const double pv_ll = -block_sum_256(accp, red);
        const double cv_ll = reuse ? kept_ll : -block_sum_256(accc, red);
        const double r = pv_prior + pv_ll - cv_prior - cv_ll;               // :29
        double u;
        if (a.U) { u = a.U[q]; q += 1; }
        else u = item_uniform(a.seed, a.iter, GPIRT_ST_BETA, item, (uint32_t)(2 * k + 1));
        if (log(u) < r) { cv[k] = pv[k]; kept_ll = pv_ll; } else { pv[k] = cv[k]; kept_ll = cv_ll; }   // :30-35
    }
    if (threadIdx.x == 0) { a.beta[0 + 2 * j] = cv[0]; a.beta[1 + 2 * j] = cv[1]; }
    if (a.mu)
        for (int64_t i = threadIdx.x; i < n; i += 256) a.mu[i + j * n] = cv[0] + a.theta[i] * cv[1];
    if (a.mu_star)
        for (int64_t i = threadIdx.x; i < a.N; i += 256)
            a.mu_star[i + j * a.N] = cv[0] + (-5.0 + (double)i * 0.01) * cv[1];
}

__global__ void linear_mean_kernel(const double* __restrict__ x, int64_t n, const double* __restrict__ beta,
                                   int64_t m, double* __restrict__ mu)
{
    const int64_t total = n * m;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = g / n, i = g - j * n;
        mu[g] = beta[0 + 2 * j] + x[i] * beta[1 + 2 * j];
    }
}

__global__ void axpy_kernel(double* __restrict__ acc, const double* __restrict__ x, int64_t count)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < count;
         g += (int64_t)gridDim.x * blockDim.x)
        acc[g] += x[g];
}

inline unsigned grid_for(int64_t total)
{
    int64_t b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

int launch_colnorm_s(hipStream_t stream, const double* tmp, int64_t n, int64_t N, int64_t ld, double* s)
{
    if (N <= 0) return 0;
    hipLaunchKernelGGL(colnorm_s_kernel, dim3((unsigned)N), dim3(256), 0, stream, tmp, n, ld, s);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_lowrank_s(hipStream_t stream, const double* V, int64_t N, int r, const double* G, int64_t ldg, double* s)
{
    if (N <= 0) return 0;
    hipLaunchKernelGGL(lowrank_s_kernel, dim3((unsigned)N), dim3(128), 0, stream, V, N, r, G, ldg, s);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_fstar_epilogue(hipStream_t stream, const FstarEpiArgs& a)
{
    if (a.N <= 0 || a.m <= 0) return 0;
    const int* off = nullptr;
    if (a.U) {
        if (!a.off_scratch) { set_error("R-stream fstar replay needs the offset scratch (N + 1 ints)"); return GPIRT_E_ARG; }
        hipLaunchKernelGGL(fstar_offsets_kernel, dim3(1), dim3(256), 0, stream, a.s, (int)a.N, a.off_scratch);
        off = a.off_scratch;
    }
    dim3 grid((unsigned)((a.N + 255) / 256), (unsigned)a.m);
    hipLaunchKernelGGL(fstar_epilogue_kernel, grid, dim3(256), 0, stream, a, off);
    if (a.U) hipLaunchKernelGGL(fstar_advance_kernel, dim3(1), dim3(1), 0, stream, a.pos, off, (int)a.N, a.m);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_indicators(hipStream_t stream, const double* y, int64_t n, int64_t m, double* Ypm)
{
    hipLaunchKernelGGL(indicators_kernel, dim3(grid_for(n * m)), dim3(256), 0, stream, y, n * m, Ypm);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_loglik_terms(hipStream_t stream, const double* fstar, int64_t N, int64_t m, double* Gpm, int64_t ldg, const int* run_if)
{
    // (a conditional launch almost never runs: a small grid keeps what it costs to find that out at a couple of microseconds)
    const unsigned blocks = run_if ? (grid_for(N * m) < 256u ? grid_for(N * m) : 256u) : grid_for(N * m);
    hipLaunchKernelGGL(loglik_terms_kernel, dim3(blocks), dim3(256), 0, stream, fstar, N, m, Gpm, ldg, run_if);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_theta_sample(hipStream_t stream, const ThetaArgs& a)
{
    if (a.n <= 0) return 0;
    if (a.N > 1024) { set_error("theta grid larger than 1024 points is not supported"); return GPIRT_E_ARG; }
    hipLaunchKernelGGL(theta_sample_kernel, dim3((unsigned)a.n), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_draw_beta(hipStream_t stream, const BetaArgs& a)
{
    if (a.m <= 0) return 0;
    hipLaunchKernelGGL(draw_beta_kernel, dim3((unsigned)a.m), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_linear_mean(hipStream_t stream, const double* x, int64_t n, const double* beta, int64_t m, double* mu)
{
    if (n * m <= 0) return 0;
    hipLaunchKernelGGL(linear_mean_kernel, dim3(grid_for(n * m)), dim3(256), 0, stream, x, n, beta, m, mu);
    GP_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, int64_t rows, int64_t cols, int64_t ldi,
                                                        double* __restrict__ out, int64_t ldo)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + tx, c = c0 + ty + 8 * k;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? in[r + c * ldi] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c0 + tx, r = r0 + ty + 8 * k;
        if (r < rows && c < cols) out[c + r * ldo] = tile[tx][ty + 8 * k];
    }
}

int launch_transpose(hipStream_t stream, const double* in, int64_t rows, int64_t cols, int64_t ldi, double* out, int64_t ldo)
{
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32)), dim3(256), 0, stream,
                       in, rows, cols, ldi, out, ldo);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_axpy_irf(hipStream_t stream, double* acc, const double* fstar, int64_t count)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(count)), dim3(256), 0, stream, acc, fstar, count);
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
