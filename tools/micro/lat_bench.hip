// instruction-latency probes for the latency-bound chain kernels (one wave, dependent chains), gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
constexpr int N = 512;
#define T_BEGIN(var) asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(var) :: "memory")
#define T_END(var, slot) do { asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(var) : "memory"); if (lane == 0) ts[slot] = t1 - t0; } while (0)
__global__ void k(double* out, long long* ts, double seed)
{
    __shared__ double sh[256];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-9, y = 1.0000001;
    unsigned long long t0, t1;
    // 0: dependent v_fma_f64 chain
    T_BEGIN(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, y, 1e-9);
    T_END(x, 0);
    // 1: 4 independent fma chains (issue rate)
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    T_BEGIN(a0);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { a0 = fma(a0, y, 1e-9); a1 = fma(a1, y, 1e-9); a2 = fma(a2, y, 1e-9); a3 = fma(a3, y, 1e-9); }
    T_END(a0, 1);
    x = a0 + a1 + a2 + a3;
    // 2: dependent raw v_rsq_f64 chain
    double r = fabs(x) + 1.0;
    T_BEGIN(r);
#pragma unroll
    for (int i = 0; i < N; ++i) r = __builtin_amdgcn_rsq(r) + 1.0;
    T_END(r, 2);
    // 3: dependent library rsqrt chain
    double q = r + 2.0;
    T_BEGIN(q);
#pragma unroll
    for (int i = 0; i < N; ++i) q = rsqrt(q) + 1.0;
    T_END(q, 3);
    // 4: readlane -> fma dependent chain
    double z = q;
    T_BEGIN(z);
#pragma unroll
    for (int i = 0; i < N; ++i) { const double l = readlane_f64(z, i & 63); z = fma(z, 0.5, l * 1e-9); }
    T_END(z, 4);
    // 5: LDS write -> read (same wave) dependent chain
    double w = z;
    T_BEGIN(w);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { sh[lane] = w; w = sh[(lane + 1) & 63] + 1e-9; }
    T_END(w, 5);
    // 6: dependent MFMA f64 16x16x4 chain (acc dependency)
    d4 acc = { w, w, w, w }; double accs = w;
    { double tmp = acc[0]; T_BEGIN(tmp); acc[0] = tmp; }
#pragma unroll
    for (int i = 0; i < N / 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
    accs = acc[0]; T_END(accs, 6);
    // 7: ds_bpermute dependent chain
    int v = (int)acc[0] + lane; 
    T_BEGIN(v);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) v = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, v) + 1;
    T_END(v, 7);
    // 8: independent readlanes (issue rate): 2 readlanes + fma into separate accumulators
    double b0 = w, b1 = w, b2 = w, b3 = w;
    T_BEGIN(b0);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        b0 = fma(b0, 0.5, readlane_f64(z, (4 * i) & 63)); b1 = fma(b1, 0.5, readlane_f64(z, (4 * i + 1) & 63));
        b2 = fma(b2, 0.5, readlane_f64(z, (4 * i + 2) & 63)); b3 = fma(b3, 0.5, readlane_f64(z, (4 * i + 3) & 63));
    }
    T_END(b0, 8);
    out[lane] = x + r + q + z + w + acc[1] + v + b0 + b1 + b2 + b3;
}
int main()
{
    double* d; long long* ts;
    CK(hipMalloc(&d, 64 * 8)); CK(hipMalloc(&ts, 16 * 8));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, ts, 1.5); CK(hipDeviceSynchronize()); }
    long long h[16]; CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
    printf("ns per op (100 MHz s_memrealtime)\n");
    printf("dep fma f64        %.1f\n", 10.0 * h[0] / (double)N);
    printf("indep fma f64 x4   %.1f\n", 10.0 * h[1] / (double)N);
    printf("dep v_rsq_f64+add  %.1f\n", 10.0 * h[2] / (double)N);
    printf("dep rsqrt()+add    %.1f\n", 10.0 * h[3] / (double)N);
    printf("dep readlane+fma   %.1f\n", 10.0 * h[4] / (double)N);
    printf("dep lds wr->rd     %.1f\n", 10.0 * h[5] / (double)(N / 4));
    printf("dep mfma f64       %.1f\n", 10.0 * h[6] / (double)(N / 4));
    printf("dep ds_bpermute    %.1f\n", 10.0 * h[7] / (double)(N / 4));
    printf("indep readlane+fma %.1f\n", 10.0 * h[8] / (double)N);
    return 0;
}
