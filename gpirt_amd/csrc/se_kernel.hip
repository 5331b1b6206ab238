// se_kernel.hip -- squared-exponential covariance K(x1, x2) (src/covariance-function.cpp:3-14)
// with the +jitter diagonal of src/gpirtMCMC.cpp:16,77,96 fused in.
//
// HBM-write bound (8 n1 n2 bytes out, 8 (n1+n2) bytes in): each work-group produces a 128-row x
// 32-column tile; a lane owns two consecutive rows, so every store instruction writes one
// contiguous 1 KiB segment of a column.  The `lower` variant visits only the 128 x 128 tiles on or
// below the block diagonal -- all the blocked Cholesky ever reads -- halving bytes and exp() work.
#include "common.h"
#include "kernels.h"

namespace gpirt {

namespace {

constexpr int TR = 128;   // tile rows
constexpr int TC = 32;    // tile columns

__device__ __forceinline__ void se_tile(const double* __restrict__ x1, int64_t n1,
                                        const double* __restrict__ x2, int64_t n2,
                                        double* __restrict__ out, int64_t ld, double jitter,
                                        int64_t i0, int64_t j0, bool lower_only, bool fp32 = false)
{
    const int t = threadIdx.x;
    const int64_t r = i0 + (t & 63) * 2;
    const bool in0 = r < n1, in1 = (r + 1) < n1;
    const double a0 = in0 ? x1[r] : 0.0;
    const double a1 = in1 ? x1[r + 1] : 0.0;
    const bool vec = in1 && ((ld & 1) == 0) && ((((uintptr_t)out) & 15) == 0);
#pragma unroll
    for (int p = 0; p < TC / 4; ++p) {
        const int64_t c = j0 + (t >> 6) + 4 * p;
        if (c >= n2) break;
        const double b = x2[c];
        const double d0 = a0 - b, d1 = a1 - b;
        double v0, v1;
        if (fp32) {      // config C5: single-precision kernel build feeding the fp64 factorisation
            v0 = (double)expf((float)(-0.5 * d0 * d0));
            v1 = (double)expf((float)(-0.5 * d1 * d1));
        } else {
            v0 = exp(-0.5 * d0 * d0);
            v1 = exp(-0.5 * d1 * d1);
        }
        if (r == c) v0 += jitter;
        if (r + 1 == c) v1 += jitter;
        if (lower_only) {                 // strict upper part of a diagonal tile: exact zeros
            if (r < c) v0 = 0.0;
            if (r + 1 < c) v1 = 0.0;
        }
        double* o = out + r + c * ld;
        if (vec) {
            *reinterpret_cast<double2*>(o) = make_double2(v0, v1);
        } else {
            if (in0) o[0] = v0;
            if (in1) o[1] = v1;
        }
    }
}

__global__ __launch_bounds__(256) void se_kernel_full(const double* __restrict__ x1, int64_t n1,
                                                      const double* __restrict__ x2, int64_t n2,
                                                      double* __restrict__ out, int64_t ld,
                                                      double jitter, int rblocks)
{
    const int64_t bi = blockIdx.x % rblocks, bj = blockIdx.x / rblocks;
    se_tile(x1, n1, x2, n2, out, ld, jitter, bi * TR, bj * TC, false);
}

// blockIdx.x enumerates (lower block pair, 32-column strip) ; pairs are 128 x 128
__global__ __launch_bounds__(256) void se_kernel_lower(const double* __restrict__ x, int64_t n,
                                                       double* __restrict__ out, int64_t ld,
                                                       double jitter, bool fp32)
{
    const int strip = blockIdx.x & 3;
    const int t = blockIdx.x >> 2;
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    const int bi = r, bj = t - r * (r + 1) / 2;
    se_tile(x, n, x, n, out, ld, jitter, (int64_t)bi * TR, (int64_t)bj * 128 + strip * TC, bi == bj, fp32);
}

}  // namespace

int launch_se_kernel(hipStream_t stream, const double* x1, int64_t n1, const double* x2, int64_t n2,
                     double* out, int64_t ld, double jitter)
{
    if (n1 <= 0 || n2 <= 0) return 0;
    const int rblocks = (int)((n1 + TR - 1) / TR);
    const int64_t cblocks = (n2 + TC - 1) / TC;
    hipLaunchKernelGGL(se_kernel_full, dim3((unsigned)(rblocks * cblocks)), dim3(256), 0, stream,
                       x1, n1, x2, n2, out, ld, jitter, rblocks);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_se_kernel_lower(hipStream_t stream, const double* x, int64_t n, double* out, int64_t ld,
                           double jitter, bool fp32)
{
    if (n <= 0) return 0;
    const int64_t nb = (n + 127) / 128;
    const int64_t pairs = nb * (nb + 1) / 2;
    hipLaunchKernelGGL(se_kernel_lower, dim3((unsigned)(pairs * 4)), dim3(256), 0, stream, x, n, out,
                       ld, jitter, fp32);
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
