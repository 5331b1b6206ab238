// Does a gfx950 LDS-DMA load (global_load_lds_dwordx4) reach every part of a 160 KB LDS allocation, and is the image
// lane-linear (wave-uniform base + 16 B x lane)?  One work-group, four waves, each wave copies 1 KiB pieces to a list of
// LDS offsets; the work-group then reads LDS back with ordinary ds_reads and writes it out.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/ldsdma_probe.hip -o tools/micro/ldsdma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(const double* src, double* out, const int* offs, int noffs, int aux_sc1)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 160 * 1024 / 8 - 64; i += 256) smem[i] = -1.0;
    __syncthreads();
    for (int q = wave; q < noffs; q += 4) {
        const int off = __builtin_amdgcn_readfirstlane(offs[q]);          // in doubles
        const double* g = src + (size_t)q * 128 + 2 * lane;               // 16 B per lane
        __attribute__((address_space(3))) double* l = (__attribute__((address_space(3))) double*)(smem + off);
        if (aux_sc1) __builtin_amdgcn_global_load_lds(g, l, 16, 0, 16);
        else __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = 0; q < noffs; ++q)
        if (threadIdx.x < 128) out[(size_t)q * 128 + threadIdx.x] = smem[offs[q] + threadIdx.x];
}

int main()
{
    std::vector<int> offs = { 0, 128, 4352, 8000, 8192 - 128, 8192, 8200, 9000, 12000, 16384, 17000, 19000, 20000 };   // doubles: 8192 = 64 KB
    const int no = (int)offs.size();
    std::vector<double> h((size_t)no * 128);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1000.0 + (double)i;
    double *ds, *dout; int* doffs;
    CK(hipMalloc(&ds, h.size() * 8)); CK(hipMalloc(&dout, h.size() * 8)); CK(hipMalloc(&doffs, no * 4));
    CK(hipMemcpy(ds, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(doffs, offs.data(), no * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    for (int aux = 0; aux < 2; ++aux) {
        CK(hipMemset(dout, 0, h.size() * 8));
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 160 * 1024 - 512, 0, ds, dout, doffs, no, aux);
        CK(hipDeviceSynchronize());
        std::vector<double> o(h.size());
        CK(hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost));
        for (int q = 0; q < no; ++q) {
            int bad = 0;
            for (int i = 0; i < 128; ++i) bad += o[(size_t)q * 128 + i] != h[(size_t)q * 128 + i];
            printf("aux %2d  LDS offset %6d B: %s (%d of 128 differ; first got %.0f want %.0f)\n", aux ? 16 : 0, offs[q] * 8, bad ? "WRONG" : "ok", bad,
                   o[(size_t)q * 128], h[(size_t)q * 128]);
        }
    }
    return 0;
}
