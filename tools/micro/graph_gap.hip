// graph_gap.hip -- what a DEPENDENT chain of small kernels costs per kernel when launched into a stream one by one and when
// replayed as a hipGraph (captured from the same stream): the R-stream predictor's phase A is such a chain (three kernels a
// pass, ~1000 per draw).   hipcc -O3 --offload-arch=gfx950 tools/micro/graph_gap.hip -o tools/micro/graph_gap && ./graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Args { unsigned long long* anchor; float* buf; int n; int spin; };
// a stand-in for a pass kernel: reads the anchor, touches a little memory, work-group 0 bumps the anchor
__global__ __launch_bounds__(256) void link_kernel(Args a)
{
    const unsigned long long v = a.anchor[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float x = (i < a.n) ? a.buf[i] : 0.0f;
    for (int q = 0; q < a.spin; ++q) x = __builtin_fmaf(x, 0.999f, 1.0f);          // (a body of a few microseconds: the host gets ahead of the device)
    if (i < a.n) a.buf[i] = x * 0.5f + (float)(v & 7);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.anchor[0] = v + 1;
}
int main(int argc, char** argv)
{
    hipStream_t st; CK(hipStreamCreate(&st));
    Args a; a.n = 288 * 256; a.spin = argc > 1 ? atoi(argv[1]) : 0;
    printf("spin %d\n", a.spin);
    CK(hipMalloc(&a.anchor, 64)); CK(hipMemset(a.anchor, 0, 64)); CK(hipMalloc(&a.buf, a.n * 4)); CK(hipMemset(a.buf, 0, a.n * 4));
    const int CHAIN = 3000, GRID = 288;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        auto h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < CHAIN; ++k) hipLaunchKernelGGL(link_kernel, dim3(GRID), dim3(256), 0, st, a);
        CK(hipEventRecord(e1, st));
        auto h1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream launches: %.2f us per kernel on the device, %.2f us of host time per launch\n", ms * 1e3 / CHAIN,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / CHAIN);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 300; ++k) hipLaunchKernelGGL(link_kernel, dim3(GRID), dim3(256), 0, st, a);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
        auto h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < CHAIN / 300; ++k) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        auto h1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph of 300 kernel nodes x %d: %.2f us per kernel on the device, %.2f us of host time per kernel\n", CHAIN / 300, ms * 1e3 / CHAIN,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / CHAIN);
    }
    unsigned long long v; CK(hipMemcpy(&v, a.anchor, 8, hipMemcpyDeviceToHost));
    printf("anchor %llu (expected %d)\n", v, 4 * CHAIN);
    return 0;
}
