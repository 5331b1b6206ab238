// micro-benchmark of the fused 256-row trsm leaf (gpirt_amd/csrc/trsm.hip) with per-stage stamps of work-group 0
#include "../../gpirt_amd/csrc/trsm.hip"
#include <vector>
#include <cmath>
#include <cstdlib>
namespace gpirt { void set_error(const char* fmt, ...) { printf("error: %s\n", fmt); }
int launch_gemm(gpirt_handle_t, hipStream_t, bool, bool, int, int64_t, int64_t, int64_t, double, const double*, int64_t, const double*, int64_t, double, double*, int64_t) { return 0; } }
using namespace gpirt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main()
{
    const int n = 256; const int64_t nrhs = 2025, ldb = 8192;
    std::vector<double> L((size_t)n * n, 0.0), B((size_t)ldb * nrhs);
    srand(5);
    for (int c = 0; c < n; ++c) for (int r = c; r < n; ++r) L[r + (size_t)c * n] = (r == c) ? 1.0 + rand() / (double)RAND_MAX : 0.1 * (rand() / (double)RAND_MAX - 0.5);
    for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
    double *dL, *dB; long long* dts;
    CK(hipMalloc(&dL, L.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dts, 32 * 8));
    CK(hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(trsm_leaf256_kernel<false>, dim3((unsigned)((nrhs + 63) / 64)), dim3(256), 0, 0, dL, (int64_t)n, n, dB, ldb, nrhs, dts);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long ts[32]; CK(hipMemcpy(ts, dts, sizeof(ts), hipMemcpyDeviceToHost));
        printf("leaf256: %.1f us; WG0 stages (us):", ms * 1e3);
        for (int k = 1; k <= 12; ++k) printf(" %.2f", (ts[k] - ts[k - 1]) / 100.0);
        printf("  total %.2f\n", (ts[12] - ts[0]) / 100.0);
    }
    return 0;
}
