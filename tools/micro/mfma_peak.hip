// sustained fp64 MFMA rate of the whole chip with nothing but MFMAs in flight (registers only): the ceiling
// the GEMM main loop is judged against (78.6 TFLOP/s nominal at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
// CH consecutive MFMAs go to the same accumulator before the next one is touched (CH = 1: round-robin)
template <int NACC, int CH>
__global__ __launch_bounds__(256) void kc(double* out, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{ 0.0, 0.0, 0.0, 0.0 };
    double a[4], b[4];
    for (int q = 0; q < 4; ++q) { a[q] = threadIdx.x * 1e-3 + q; b[q] = blockIdx.x * 1e-3 + q; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c & 3], b[(c + i) & 3], acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The same with the MFMAs written as inline assembly, accumulators pinned in VGPRs: the compiler's versions above
// shuffle accumulators between AGPRs and VGPRs around every visit (v_accvgpr moves), which is what their
// "chain length" dependence measures.  This one is the honest ceiling for a stream of MFMAs.
template <int NACC, int CH>
__global__ __launch_bounds__(256) void ka(double* out, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{ 0.0, 0.0, 0.0, 0.0 };
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{ 0.0, 0.0, 0.0, 0.0 };
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    double* d; CK(hipMalloc(&d, 4096 * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // (a) how the rate scales with the number of busy CUs (one work-group = 4 waves = one per SIMD)
    for (int grid : { 16, 64, 128, 256 }) {
        const int iters = 25000;
        hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, d, 10);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, d, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * 4 * iters * 16.0 * 2048.0;
        printf("16 acc, %3d work-groups, %7.2f ms: %6.1f TFLOP/s, %.1f ns per MFMA per wave\n", grid, ms, flops / ms / 1e9, ms * 1e6 / (iters * 16.0));
    }
    // (c) 16 accumulators, CH consecutive MFMAs per accumulator (dependent: the accumulator is forwarded)
    {
        const int grid = 256, iters = 6000;
        auto run = [&](auto kern, int ch, const char* name) -> int {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 10);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (double)grid * 4 * iters * 16.0 * ch * 2048.0;
            printf("%s: %7.2f ms: %6.1f TFLOP/s, %.1f ns per MFMA per wave\n", name, ms, flops / ms / 1e9, ms * 1e6 / (iters * 16.0 * ch));
            return 0;
        };
        if (run(kc<16, 1>, 1, "16 acc x 1 in a row")) return 1;
        if (run(kc<16, 2>, 2, "16 acc x 2 in a row")) return 1;
        if (run(kc<16, 4>, 4, "16 acc x 4 in a row")) return 1;
        if (run(kc<16, 8>, 8, "16 acc x 8 in a row")) return 1;
        if (run(kc<16, 16>, 16, "16 acc x 16 in a row")) return 1;
    }
    // (c2) inline-assembly MFMAs, accumulators pinned
    {
        const int iters = 6000;
        auto run3 = [&](auto kern, int grid, int nacc, int ch, const char* name) -> int {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 10);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (double)grid * 4 * iters * (double)nacc * ch * 2048.0;
            printf("asm, %s, %d work-groups: %7.2f ms: %6.1f TFLOP/s\n", name, grid, ms, flops / ms / 1e9);
            return 0;
        };
        if (run3(ka<16, 1>, 256, 16, 1, "16 acc x 1")) return 1;
        if (run3(ka<16, 4>, 256, 16, 4, "16 acc x 4")) return 1;
        if (run3(ka<16, 16>, 256, 16, 16, "16 acc x 16")) return 1;
        if (run3(ka<16, 1>, 512, 16, 1, "16 acc x 1")) return 1;
        if (run3(ka<8, 1>, 512, 8, 1, " 8 acc x 1")) return 1;
        if (run3(ka<4, 1>, 512, 4, 1, " 4 acc x 1")) return 1;
        if (run3(ka<2, 1>, 512, 2, 1, " 2 acc x 1")) return 1;
        if (run3(ka<1, 1>, 512, 1, 1, " 1 acc x 1")) return 1;
    }
    // (d) the same with two waves per SIMD (8 accumulators per wave, 512 work-groups)
    {
        const int grid = 512, iters = 6000;
        auto run2 = [&](auto kern, int ch, const char* name) -> int {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 10);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (double)grid * 4 * iters * 8.0 * ch * 2048.0;
            printf("2 waves/SIMD, %s: %7.2f ms: %6.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", name, ms, flops / ms / 1e9,
                   ms * 1e6 / (iters * 8.0 * ch * 2.0));
            return 0;
        };
        if (run2(kc<8, 1>, 1, "8 acc x 1 in a row")) return 1;
        if (run2(kc<8, 4>, 4, "8 acc x 4 in a row")) return 1;
        if (run2(kc<8, 8>, 8, "8 acc x 8 in a row")) return 1;
        if (run2(kc<8, 16>, 16, "8 acc x 16 in a row")) return 1;
    }
    // (b) two waves per SIMD (4 accumulators each, so both fit in the register file)
    for (int grid : { 256, 512, 1024 }) {
        const int iters = 100000;
        hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, 10);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * 4 * iters * 4.0 * 2048.0;
        printf(" 4 acc, %4d work-groups, %7.2f ms: %6.1f TFLOP/s\n", grid, ms, flops / ms / 1e9);
    }
    return 0;
}
