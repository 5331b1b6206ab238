// micro-benchmark of the split sub-panel (gpirt_amd/csrc/panel.hip): the chain launch on the diagonal owners + the rows
// just below them, and the rest of the rows on panel_rows_kernel (32-row work-groups) -- beside the chain launch on a
// second stream, or after it -- against the one-launch sub-panel, bit for bit.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/micro/rows_bench.hip -o tools/micro/rows_bench
//   usage: rows_bench [n = 8192] [W = 512] [window rows = 1024]
#include "../../gpirt_amd/csrc/panel.hip"
#include <vector>
#include <cmath>
#include <cstdlib>
#include <cstring>
namespace gpirt { void set_error(const char* fmt, ...) { printf("error: %s\n", fmt); } }
using namespace gpirt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 8192;
    const int64_t W = argc > 2 ? atoll(argv[2]) : 512;
    const int64_t win = argc > 3 ? atoll(argv[3]) : 1024;
    std::vector<double> th(n), S((size_t)n * W);
    srand(3);
    for (auto& v : th) { double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0); v = sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2); }
    for (int64_t c = 0; c < W; ++c)
        for (int64_t r = 0; r < n; ++r) S[r + c * n] = exp(-0.5 * (th[r] - th[c]) * (th[r] - th[c])) + (r == c ? 1e-3 : 0.0);
    double* dA; CK(hipMalloc(&dA, (size_t)n * W * 8));
    gpirt_handle_s h;
    { hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); h.n_cu = prop.multiProcessorCount; }
    CK(hipMalloc(&h.d_info, 64)); CK(hipMemset(h.d_info, 0, 64)); CK(hipDeviceSynchronize());
    hipStream_t s1, s2;
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    h.stream = s1;
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    std::vector<double> Lref((size_t)n * W), L((size_t)n * W);
    auto reset = [&]() { return hipMemcpy(dA, S.data(), S.size() * 8, hipMemcpyHostToDevice); };
    // ---- reference: one launch for all rows
    for (int rep = 0; rep < 3; ++rep) {
        CK(reset());
        CK(hipEventRecord(e0, s1));
        if (launch_panel_ll(&h, s1, dA, n, n, 0, W)) return 1;
        CK(hipEventRecord(e1, s1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("one launch, all %lld rows: %.1f us\n", (long long)n, ms * 1e3);
    }
    CK(hipMemcpy(Lref.data(), dA, Lref.size() * 8, hipMemcpyDeviceToHost));
    auto compare = [&](const char* tag) {
        hipMemcpy(L.data(), dA, L.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; double worst = 0;
        for (int64_t c = 0; c < W; ++c)
            for (int64_t r = c; r < n; ++r) {
                const double a = L[r + c * n], b = Lref[r + c * n];
                if (memcmp(&a, &b, 8) != 0) { ++bad; worst = fmax(worst, fabs(a - b)); }
            }
        printf("  %s: %zu elements differ from the one-launch factor (max |diff| %.3e)\n", tag, bad, worst);
    };
    for (int lean = 1; lean >= 0; --lean) {
        for (int mode = 0; mode < 2; ++mode) {          // 0: beside the chain launch, 1: after it
            for (int rep = 0; rep < 3; ++rep) {
                CK(reset());
                unsigned long long epoch = 0;
                CK(hipEventRecord(e0, s1));
                if (launch_panel_ll(&h, s1, dA, n, n, 0, W, win, &epoch)) return 1;
                CK(hipEventRecord(e1, s1));
                if (mode == 1) CK(hipStreamWaitEvent(s2, e1, 0));
                CK(hipEventRecord(f0, s2));
                if (launch_panel_rows(&h, s2, dA, n, n, 0, W, win, n, epoch, lean != 0)) return 1;
                CK(hipEventRecord(f1, s2));
                CK(hipDeviceSynchronize());
                float a, b, c; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, f0, f1)); CK(hipEventElapsedTime(&c, e0, f1));
                printf("%s rows kernel %s the chain launch (window %lld rows): chain %.1f us, rows %.1f us, both done after %.1f us\n",
                       lean ? "lean (32-row)" : "whole-CU", mode ? "AFTER" : "BESIDE", (long long)win, a * 1e3, b * 1e3, c * 1e3);
            }
            compare(lean ? (mode ? "lean/after" : "lean/beside") : (mode ? "fat/after" : "fat/beside"));
        }
    }
    int info[8]; CK(hipMemcpy(info, h.d_info, 32, hipMemcpyDeviceToHost));
    printf("info %d guard %d\n", info[0], info[1]);
    return 0;
}
