// hand-off latency between two work-groups of one launch: a 32 KB block (a 64 x 64 fp64 L_jj) + a flag, the way
// panel.hip hands L_jj along the pivot chain (sc1 stores, vmcnt(0), barrier, sc1 flag; sc1 poll, barrier, sc1 loads),
// with the consumer on the SAME XCD as the producer (blockIdx 8) or on another one (blockIdx 1).  Round trips timed
// in-kernel (100 MHz wall clock); every word of the payload is checked.   usage: handoff_bench [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ unsigned long long ld_flag(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: sc1 stores / sc1 loads (the shipped protocol); 1: PLAIN stores / sc1 loads (valid only when both share an L2:
// timing experiment for the same-XCD case, the check tells whether it was); 2: flag only (no payload)
template <int MODE>
__global__ __launch_bounds__(256) void k(double* buf, unsigned long long* flags, long long* out, int consumer, int iters)
{
    extern __shared__ double lds[];        // (100 KB: one work-group per CU)
    __shared__ unsigned long long seen;
    const int t = threadIdx.x, b = blockIdx.x;
    if (b != 0 && b != consumer) return;
    if (t == 0) out[8 + (b != 0)] = xcc_id();
    unsigned long long* f_go = flags;          // producer -> consumer
    unsigned long long* f_back = flags + 32;   // consumer -> producer (another line)
    long long bad = 0, t_total = 0;
    for (int it = 1; it <= iters; ++it) {
        if (b == 0) {
            const long long t0 = wall_clock64();
            if (MODE != 2) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const double v = (double)(it * 4096 + t + 256 * q);
                    if (MODE == 0) __hip_atomic_store(buf + t + 256 * q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else buf[t + 256 * q] = v;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (t == 0) {
                __hip_atomic_store(f_go, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (ld_flag(f_back) < (unsigned long long)it && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                t_total += wall_clock64() - t0;
            }
            __syncthreads();
        } else {
            if (t == 0) {
                int spins = 0;
                while (ld_flag(f_go) < (unsigned long long)it && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
                seen = 1;
            }
            __syncthreads();
            double s = 0.0;
            if (MODE != 2) {
                double v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = __hip_atomic_load(buf + t + 256 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int q = 0; q < 16; ++q) { if (v[q] != (double)(it * 4096 + t + 256 * q)) ++bad; s += v[q]; }
                lds[t] = s;            // (the block lands in LDS in panel.hip; one store stands in for it)
            }
            __syncthreads();
            if (t == 0) __hip_atomic_store(f_back, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (b == 0 && t == 0) out[0] = t_total;
    if (b != 0) atomicAdd((unsigned long long*)&out[1], (unsigned long long)bad);
}

template <int MODE>
static int run(const char* name, int consumer, int iters, double* buf, unsigned long long* flags, long long* out)
{
    CK(hipMemset(flags, 0, 512)); CK(hipMemset(out, 0, 128)); CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipLaunchKernelGGL(k<MODE>, dim3(16), dim3(256), 100 * 1024, 0, buf, flags, out, consumer, iters);
    CK(hipDeviceSynchronize());
    long long h[16]; CK(hipMemcpy(h, out, 128, hipMemcpyDeviceToHost));
    printf("%-44s consumer block %2d  XCC %lld -> %lld  round trip %.2f us  (one way ~%.2f)  wrong words %lld\n", name, consumer, h[8], h[9],
           h[0] / 100.0 / iters, h[0] / 200.0 / iters, h[1]);
    return 0;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    double* buf; unsigned long long* flags; long long* out;
    CK(hipMalloc(&buf, 4096 * 8)); CK(hipMalloc(&flags, 512)); CK(hipMalloc(&out, 128));
    for (int rep = 0; rep < 2; ++rep)
        for (int c : { 8, 1, 2, 9 }) {
            if (run<2>("flag only (sc1 store, sc1 poll)", c, iters, buf, flags, out)) return 1;
            if (run<0>("32 KB sc1 stores + flag, sc1 loads", c, iters, buf, flags, out)) return 1;
            if (run<1>("32 KB PLAIN stores + flag, sc1 loads", c, iters, buf, flags, out)) return 1;
        }
    return 0;
}
