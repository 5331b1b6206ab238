// micro-benchmark of the persistent panel kernel (gpirt_amd/csrc/panel.hip): one panel of W columns of an
// n x n SE-kernel matrix, with per-step time stamps of selected row blocks.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/micro/panel_bench.hip -o tools/micro/panel_bench
#include "../../gpirt_amd/csrc/panel.hip"
#include <vector>
#include <cmath>
#include <cstdlib>
namespace gpirt { void set_error(const char* fmt, ...) { printf("error: %s\n", fmt); } }
using namespace gpirt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 8192;
    const int64_t W = argc > 2 ? atoll(argv[2]) : 1024;
    std::vector<double> th(n), S((size_t)n * W);
    srand(3);
    for (auto& v : th) { double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0); v = sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2); }
    for (int64_t c = 0; c < W; ++c)
        for (int64_t r = 0; r < n; ++r) S[r + c * n] = exp(-0.5 * (th[r] - th[c]) * (th[r] - th[c])) + (r == c ? 1e-3 : 0.0);
    double* dA; CK(hipMalloc(&dA, (size_t)n * n * 8));
    gpirt_handle_s h;
    { hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); h.n_cu = prop.multiProcessorCount; }
    CK(hipMalloc(&h.d_info, 64)); CK(hipMemset(h.d_info, 0, 64)); CK(hipDeviceSynchronize());
    const int nrb = (int)((n + 63) / 64);
    CK(hipMalloc(&h.panel_trace, (size_t)nrb * 40 * 8 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(dA, S.data(), S.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(h.panel_trace, 0, (size_t)nrb * 40 * 8 * 8));
        CK(hipEventRecord(e0, 0));
        if (launch_panel_ll(&h, 0, dA, n, n, 0, W)) return 1;
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("panel n=%lld W=%lld: %.1f us\n", (long long)n, (long long)W, ms * 1e3);
    }
    int info[2]; CK(hipMemcpy(info, h.d_info, 8, hipMemcpyDeviceToHost));
    printf("info %d guard %d\n", info[0], info[1]);
    std::vector<long long> tr((size_t)nrb * 40 * 8);
    CK(hipMemcpy(tr.data(), h.panel_trace, tr.size() * 8, hipMemcpyDeviceToHost));
    const int ncb = (int)(W / 64);
    long long t0 = tr[(0 * 40 + 39) * 8 + 0];
    auto us = [&](long long t) { return t ? (t - t0) / 100.0 : -1.0; };
    printf("diagonal owners: row block, [potf2 start, potf2 published] (us since owner 0 began)\n");
    for (int R = 0; R < ncb; ++R) {
        const long long* q = &tr[(R * 40 + 39) * 8];
        printf("  R=%2d  potf2 phase %8.2f .. %8.2f  [D stored +%.2f, potf2 body +%.2f, published +%.2f]", R, us(q[0]), us(q[1]),
               (q[2] - q[0]) / 100.0, (q[3] - q[2]) / 100.0, (q[1] - q[3]) / 100.0);
        if (R > 0) {
            const long long* s = &tr[(R * 40 + R - 1) * 8];
            // progressive hand-off (GPIRT_PANEL_COLS != 0): slot 2 = first column block seen, 4 = last one seen,
            // 5 = last X block exchanged, 6 = D updated; otherwise: 2 = L seen, 4 = staged + inverted, 5 = solved, 6 = stored
            printf("  last step: began %.2f, gemm done %.2f, L seen %.2f [+%.2f, +%.2f, +%.2f, +%.2f]", us(s[0]), us(s[1]), us(s[2]),
                   (s[4] - s[2]) / 100.0, (s[5] - s[4]) / 100.0, (s[6] - s[5]) / 100.0, (s[3] - s[6]) / 100.0);
        }
        printf("\n");
    }
    const int show[3] = { ncb - 1, ncb, nrb - 1 };
    for (int R : { 1, 2, 5 }) {
        const long long* d = &tr[((size_t)R * 40 + 30) * 8];
        printf("potf2 of R=%d, per block column [pivots, write-back, barrier+U1+stores]:", R);
        for (int b = 0; b < 4; ++b) printf("  [%.2f %.2f %.2f]", (d[4 * b + 1] - d[4 * b]) / 100.0, (d[4 * b + 2] - d[4 * b + 1]) / 100.0, (d[4 * b + 3] - d[4 * b + 2]) / 100.0);
        printf("\n");
    }
    for (int R : show) {
        if (R < 0 || R >= nrb) continue;
        printf("row block %d: step j: start, gemm done, L_jj seen, step done\n", R);
        for (int j = 0; j < ncb && j < (R < ncb ? R : ncb); ++j)
        {
            const long long* q = &tr[(R * 40 + j) * 8];
            printf("   j=%2d  %8.2f %8.2f %8.2f %8.2f   [after L seen: staged +%.2f, solved +%.2f, stored / X in LDS +%.2f, D updated + published +%.2f]\n", j, us(q[0]), us(q[1]),
                   us(q[2]), us(q[3]), (q[4] - q[2]) / 100.0, (q[5] - q[4]) / 100.0, (q[6] - q[5]) / 100.0, (q[3] - q[6]) / 100.0);
        }
    }
#ifdef PANEL_CHUNK_PROF
    for (int R : show) {
        if (R < 0 || R >= nrb) continue;
        printf("row block %d chunk phases per step (us summed over the step's chunks): [lds store+vmcnt wait, wait_prog+barrier, loads A, mfma A, loads B, mfma B, loop back]\n", R);
        for (int j = 1; j < ncb && j < (R < ncb ? R : ncb); ++j) {
            const long long* c = &tr[(R * 40 + 16 + j) * 8];
            printf("   j=%2d (%d chunks)  %.2f %.2f | %.2f %.2f | %.2f %.2f | %.2f   [mfma A: %lld shader cycles = %.0f per MFMA, clock %.0f MHz]\n", j, j, c[0] / 100.0, c[1] / 100.0, c[5] / 100.0, c[2] / 100.0, c[6] / 100.0, c[3] / 100.0, c[4] / 100.0, c[7], c[7] / (32.0 * j), c[2] ? 100.0 * c[7] / c[2] : 0.0);
        }
    }
#endif
    // residual of the leading W x W block
    std::vector<double> L((size_t)n * W);
    CK(hipMemcpy(L.data(), dA, L.size() * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int64_t r = 0; r < n; r += 37)
        for (int64_t c = 0; c < W && c <= r; c += 11) {
            double s = 0;
            for (int64_t k = 0; k <= c; ++k) s += L[r + k * n] * L[c + k * n];
            err = fmax(err, fabs(s - S[r + c * n]));
        }
    printf("sampled residual %.3e\n", err);
    return 0;
}
