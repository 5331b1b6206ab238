// micro-benchmark: variants of the 64-column panel solve X * L^T = A (one lane per row)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <cmath>
constexpr int NBI = 64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(double* __restrict__ A, int64_t lda, int64_t n, int64_t k0, int64_t r0,
                                         const double* __restrict__ Lc)
{
    __shared__ __attribute__((aligned(16))) double sLt[NBI * NBI];
    const int t = threadIdx.x;
    if (V != 1) {
        for (int idx = t; idx < NBI * NBI; idx += 256) sLt[idx] = Lc[idx];
        __syncthreads();
    }
    const int64_t r = r0 + (int64_t)blockIdx.x * 256 + t;
    if (r >= n) return;
    double* base = A + k0 * lda;
    double x[NBI];
#pragma unroll
    for (int c = 0; c < NBI; ++c) x[c] = (base + c * lda)[r];
    if (V == 0 || V == 1) {
#pragma unroll
        for (int c = 0; c < NBI; ++c) {
            const double* col = (V == 0) ? (sLt + c * NBI) : (Lc + c * NBI);
            const double xc = x[c] / col[c];
            x[c] = xc;
#pragma unroll
            for (int c2 = c + 1; c2 < NBI; ++c2) x[c2] = fma(-xc, col[c2], x[c2]);
        }
    } else if (V == 2) {
        // batch the LDS reads of a 16-wide column strip ahead of the FMAs that use them
#pragma unroll
        for (int c = 0; c < NBI; ++c) {
            const double* col = sLt + c * NBI;
            const double xc = x[c] / col[c];
            x[c] = xc;
#pragma unroll
            for (int s0 = (c + 1) & ~15; s0 < NBI; s0 += 16) {
                double l[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) l[q] = col[s0 + q];
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (s0 + q > c) x[s0 + q] = fma(-xc, l[q], x[s0 + q]);
            }
        }
    } else if (V == 3) {
        // 16 x 16 diagonal sub-blocks per lane, off-diagonal strips with all reads issued first
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int c = 16 * b; c < 16 * b + 16; ++c) {
                const double* col = sLt + c * NBI;
                const double xc = x[c] / col[c];
                x[c] = xc;
#pragma unroll
                for (int c2 = c + 1; c2 < 16 * b + 16; ++c2) x[c2] = fma(-xc, col[c2], x[c2]);
            }
            // strip update: x[c2] -= sum_{c in block b} x[c] * L[c2][c], c2 >= 16(b+1)
#pragma unroll
            for (int c2 = 16 * (b + 1); c2 < NBI; ++c2) {
                double acc = x[c2];
#pragma unroll
                for (int c = 16 * b; c < 16 * b + 16; ++c) acc = fma(-x[c], sLt[c * NBI + c2], acc);
                x[c2] = acc;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NBI; ++c) (base + c * lda)[r] = x[c];
}

int main()
{
    const int64_t n = 8192, lda = n;
    std::vector<double> hA((size_t)n * 64), hL(64 * 64, 0.0);
    for (int c = 0; c < 64; ++c) for (int r = c; r < 64; ++r) hL[c * 64 + r] = (r == c) ? 2.0 + 0.01 * c : 0.01 * std::sin(r * 3 + c);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = std::sin(0.001 * i);
    double *dA, *dL, *d0;
    CK(hipMalloc(&dA, sizeof(double) * n * 64)); CK(hipMalloc(&d0, sizeof(double) * n * 64)); CK(hipMalloc(&dL, sizeof(double) * 4096));
    CK(hipMemcpy(d0, hA.data(), sizeof(double) * n * 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(dL, hL.data(), sizeof(double) * 4096, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> ref, out((size_t)n * 64);
    for (int v = 0; v < 4; ++v) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemcpy(dA, d0, sizeof(double) * n * 64, hipMemcpyDeviceToDevice));
            CK(hipEventRecord(e0));
            dim3 g((n + 255) / 256), b(256);
            if (v == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, dA, lda, n, 0, 0, dL);
            if (v == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, dA, lda, n, 0, 0, dL);
            if (v == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, dA, lda, n, 0, 0, dL);
            if (v == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, dA, lda, n, 0, 0, dL);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipMemcpy(out.data(), dA, sizeof(double) * n * 64, hipMemcpyDeviceToHost));
        if (v == 0) ref = out;
        double md = 0; for (size_t i = 0; i < out.size(); ++i) md = fmax(md, fabs(out[i] - ref[i]));
        printf("variant %d: %.1f us  maxdiff vs v0 %.3e\n", v, best * 1e3, md);
    }
    return 0;
}
