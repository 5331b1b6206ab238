// Do CU-masked streams partition the chip the way a scheduler would need?  (DESIGN_HISTORY.md section 10)
//   hog:   a long kernel (many 256-thread work-groups, ~100 us each, fills every CU it may use) on a stream whose CU
//          mask excludes R CUs (every 8th... see mask below);
//   probe: 32 work-groups that each need a WHOLE CU (150 KB of LDS), launched on an unmasked high-priority stream
//          while the hog is running.  If the mask holds, the probe starts at once on the excluded CUs.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void hog(double* out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.000001;
    for (int i = 0; i < iters; ++i) a = fma(a, b, 1e-9);
    if (a == 123.456) out[0] = a;
}
__global__ void probe(long long* stamps, int iters)
{
    extern __shared__ double big[];
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = wall_clock64();
    double a = threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) a = fma(a, 1.000001, 1e-9);
    big[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = wall_clock64() + (big[1] == 77.0 ? 1 : 0);
}
int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("CUs %d\n", ncu);
    std::vector<uint32_t> mask((ncu + 31) / 32, 0xffffffffu);
    int excluded = 0;
    for (int cu = 0; cu < ncu; ++cu) if ((cu % 8) == 0) { mask[cu / 32] &= ~(1u << (cu % 32)); ++excluded; }   // every 8th CU free
    hipStream_t masked, plain, hi;
    CK(hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask.size(), mask.data()));
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    int lo_p, hi_p; CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, hi_p));
    double* out; CK(hipMalloc(&out, 8));
    long long *st, *hst; CK(hipMalloc(&st, 64 * 16)); CK(hipHostMalloc(&hst, 64 * 16, hipHostMallocDefault));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipEvent_t e0, e1, p0, p1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&p0)); CK(hipEventCreate(&p1));
    const int hog_iters = 60000, hog_wgs = ncu * 8 * 6;        // ~6 rounds of 8 work-groups per CU
    for (int mode = 0; mode < 3; ++mode) {     // 0: hog on a plain stream, 1: hog on the masked stream, 2: no hog
        hipStream_t hs = mode == 1 ? masked : plain;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, hs));
            if (mode != 2) hipLaunchKernelGGL(hog, dim3(hog_wgs), dim3(256), 0, hs, out, hog_iters);
            CK(hipEventRecord(e1, hs));
            // let the hog fill the chip, then launch the probe
            for (volatile int spin = 0; spin < 2000000; ++spin) {}
            CK(hipEventRecord(p0, hi));
            hipLaunchKernelGGL(probe, dim3(32), dim3(256), 150 * 1024, hi, st, 20000);
            CK(hipEventRecord(p1, hi));
            CK(hipDeviceSynchronize());
            float hms, pms; CK(hipEventElapsedTime(&hms, e0, e1)); CK(hipEventElapsedTime(&pms, p0, p1));
            printf("mode %d (%s): hog %.1f us, probe (32 whole-CU work-groups, ~%d us of work) took %.1f us\n", mode,
                   mode == 0 ? "hog unmasked" : (mode == 1 ? "hog masked, 1/8 of the CUs excluded" : "probe alone"),
                   hms * 1e3, 20000 * 4 / 2400 * 2, pms * 1e3);
        }
    }
    printf("excluded CUs %d\n", excluded);
    return 0;
}
