// micro-benchmark: (1) in-kernel latency of the 64 x 64 diagonal-block Cholesky, (2) flag ping-pong latency
// between two work-groups with agent-scope release/acquire.  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include "../../gpirt_amd/csrc/common.h"
#include "../../gpirt_amd/csrc/potf2.h"
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
namespace gpirt { void set_error(const char*, ...) {} }
using namespace gpirt;

__global__ __launch_bounds__(256) void k_cur(double* A, int64_t lda, int* info, long long* ts, int reps)
{
    __shared__ __attribute__((aligned(16))) double sP[2 * 4 * 64];
    __shared__ int sfail;
    for (int r = 0; r < reps; ++r) {
        double* Ar = A + (int64_t)r * 64 * lda;
        long long t0 = wall_clock64();
        potf2_64_body(Ar, lda, 64, 0, info, sP, &sfail);
        __syncthreads();
        long long t1 = wall_clock64();
        if (threadIdx.x == 0) ts[r] = t1 - t0;
    }
}

__global__ __launch_bounds__(256) void k_lds(double* A, int64_t lda, int* info, long long* ts, int reps)
{
    constexpr int LS = 68;
    __shared__ __attribute__((aligned(16))) double sD[64 * LS];
    __shared__ __attribute__((aligned(16))) double sXT[64 * POTF2_XS];
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int r = 0; r < reps; ++r) {
        double* Ar = A + (int64_t)r * 64 * lda;
        for (int idx = threadIdx.x; idx < 4096; idx += 256) sD[(idx >> 6) * LS + (idx & 63)] = Ar[(idx & 63) + (idx >> 6) * lda];
        __syncthreads();
        long long t0 = wall_clock64();
        potf2_64_lds<LS>(sD, sXT, Ar, lda, 64, 0, info, nullptr, 0, ts + 16);
        __syncthreads();
        long long t1 = wall_clock64();
        if (threadIdx.x == 0) ts[r] = t1 - t0;
    }
    if (threadIdx.x == 0) { ts[40] = wall_clock64() - w0; ts[41] = clock64() - c0; }   // shader clock against the 100 MHz wall clock
}

// the same with the progressive hand-off (COLS): inverse wave, per-column stores + counters
__global__ __launch_bounds__(256) void k_cols(double* A, int64_t lda, int* info, long long* ts, int reps,
                                              unsigned long long* flags, double* winv)
{
    constexpr int LS = 68;
    __shared__ __attribute__((aligned(16))) double sD[64 * LS];
    __shared__ __attribute__((aligned(16))) double sXT[64 * POTF2_XS];
    __shared__ __attribute__((aligned(16))) double sW[4 * 256];
    for (int r = 0; r < reps; ++r) {
        double* Ar = A + (int64_t)r * 64 * lda;
        for (int idx = threadIdx.x; idx < 4096; idx += 256) sD[(idx >> 6) * LS + (idx & 63)] = Ar[(idx & 63) + (idx >> 6) * lda];
        if (threadIdx.x < 64) sXT[threadIdx.x * POTF2_XS + 16] = 0.0;
        __syncthreads();
        long long t0 = wall_clock64();
        potf2_64_lds<LS, true>(sD, sXT, Ar, lda, 64, 0, info, nullptr, 0, ts + 16, flags + 1, 64ull * r, winv, sW);
        __syncthreads();
        long long t1 = wall_clock64();
        if (threadIdx.x == 0) ts[r] = t1 - t0;
    }
}

__global__ void k_pingpong(int* flags, double* data, long long* ts, int iters)
{
    // block 0 and block 1 alternate: publish data + flag (release), the other spins (acquire) and checks the data
    const int me = blockIdx.x, other = 1 - me;
    long long t0 = wall_clock64();
    int bad = 0;
    for (int it = 1; it <= iters; ++it) {
        if ((it & 1) == me) {
            data[threadIdx.x] = (double)it;
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&flags[0], it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0) {
                int spins = 0;
                while (__hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < it && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            if (data[threadIdx.x] != (double)it) ++bad;
        }
        __syncthreads();
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { ts[me] = t1 - t0; ts[2 + me] = bad; }
    (void)other;
}

int main()
{
    const int reps = 8, lda = 64;
    std::vector<double> M(64 * 64), S(64 * 64 * reps);
    srand(1);
    for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double s = (i == j) ? 1.0 : 0.0;
                for (int k = 0; k < 64; ++k) s += M[i + 64 * k] * M[j + 64 * k];
                S[r * 4096 + i + 64 * j] = s;
            }
    double* dA; int* dinfo; long long* dts;
    CK(hipMalloc(&dA, S.size() * 8)); CK(hipMalloc(&dinfo, 4)); CK(hipMalloc(&dts, 64 * 8));
    CK(hipMemset(dinfo, 0, 4));
    for (int pass = 0; pass < 2; ++pass) {
        CK(hipMemcpy(dA, S.data(), S.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cur, dim3(1), dim3(256), 0, 0, dA, (int64_t)lda, dinfo, dts, reps);
        CK(hipDeviceSynchronize());
        long long ts[64];
        CK(hipMemcpy(ts, dts, sizeof(ts), hipMemcpyDeviceToHost));
        printf("potf2_64 current, in-kernel (100 MHz ticks -> us):");
        for (int r = 0; r < reps; ++r) printf(" %.2f", ts[r] / 100.0);
        printf("\n");
    }
    for (int pass = 0; pass < 2; ++pass) {
        CK(hipMemcpy(dA, S.data(), S.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), 0, 0, dA, (int64_t)lda, dinfo, dts, reps);
        CK(hipDeviceSynchronize());
        long long ts[64];
        CK(hipMemcpy(ts, dts, sizeof(ts), hipMemcpyDeviceToHost));
        printf("potf2_64_lds, in-kernel (us):");
        for (int r = 0; r < reps; ++r) printf(" %.2f", ts[r] / 100.0);
        printf("\n   shader clock during the kernel: %.0f MHz", 100.0 * (double)ts[41] / (double)ts[40]);
        printf("\n   per block column [pivots, write-back, barrier+U1]:");
        for (int b = 0; b < 4; ++b) printf("  [%.2f %.2f %.2f]", (ts[16 + 4 * b + 1] - ts[16 + 4 * b]) / 100.0, (ts[16 + 4 * b + 2] - ts[16 + 4 * b + 1]) / 100.0, (ts[16 + 4 * b + 3] - ts[16 + 4 * b + 2]) / 100.0);
        printf("\n");
    }
    {
        unsigned long long* dfl; double* dw;
        CK(hipMalloc(&dfl, 64)); CK(hipMemset(dfl, 0, 64)); CK(hipMalloc(&dw, 1024 * 8));
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipMemcpy(dA, S.data(), S.size() * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_cols, dim3(1), dim3(256), 0, 0, dA, (int64_t)lda, dinfo, dts, reps, dfl, dw);
            CK(hipDeviceSynchronize());
            long long ts[64];
            CK(hipMemcpy(ts, dts, sizeof(ts), hipMemcpyDeviceToHost));
            printf("potf2_64_lds<COLS>, in-kernel (us):");
            for (int r = 0; r < reps; ++r) printf(" %.2f", ts[r] / 100.0);
            printf("\n   per block column [pivots, write-back, barrier+U1+stores]:");
            for (int b = 0; b < 4; ++b) printf("  [%.2f %.2f %.2f]", (ts[16 + 4 * b + 1] - ts[16 + 4 * b]) / 100.0, (ts[16 + 4 * b + 2] - ts[16 + 4 * b + 1]) / 100.0, (ts[16 + 4 * b + 3] - ts[16 + 4 * b + 2]) / 100.0);
            printf("\n");
        }
        // W_3 of the last block against L
        std::vector<double> W(1024), Lh(64 * 64);
        CK(hipMemcpy(W.data(), dw, 1024 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(Lh.data(), dA + (size_t)(reps - 1) * 64 * lda, 64 * 64 * 8, hipMemcpyDeviceToHost));
        double werr = 0;
        for (int b = 0; b < 4; ++b)
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double sacc = 0;
                    for (int k = 0; k < 16; ++k) sacc += ((i >= k) ? Lh[(16 * b + i) + 64 * (16 * b + k)] : 0.0) * W[256 * b + j * 16 + k];
                    werr = fmax(werr, fabs(sacc - (i == j ? 1.0 : 0.0)));
                }
        printf("max |L_bb W_b - I| = %.3e\n", werr);
    }
    // residual of the last block
    std::vector<double> Lh(S.size());
    CK(hipMemcpy(Lh.data(), dA, S.size() * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int k = 0; k <= j; ++k) s += Lh[i + 64 * k] * Lh[j + 64 * k];
            err = fmax(err, fabs(s - S[i + 64 * j]));
        }
    printf("residual %.3e\n", err);
    int* dflags; double* ddata;
    CK(hipMalloc(&dflags, 64)); CK(hipMalloc(&ddata, 256 * 8)); CK(hipMemset(dflags, 0, 64));
    const int iters = 2000;
    // 2 blocks land on different XCDs (round-robin dispatch)
    hipLaunchKernelGGL(k_pingpong, dim3(2), dim3(256), 0, 0, dflags, ddata, dts, iters);
    CK(hipDeviceSynchronize());
    long long ts[4];
    CK(hipMemcpy(ts, dts, sizeof(ts), hipMemcpyDeviceToHost));
    printf("ping-pong: %.3f us per hand-off (bad data reads: %lld %lld)\n", ts[0] / 100.0 / iters, ts[2], ts[3]);
    return 0;
}
