// flagsync.h -- progress counters between the work-groups of ONE persistent kernel (panel.hip) through device memory.
//
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms", the sc1 row for one work-group per CU and hipMalloc memory;
// GPIRT_PANEL_FENCES=2 at build time keeps the fenced form this replaced):
//   producer: EVERY handed-off byte is stored sc1 (write-through at agent scope: __hip_atomic_store relaxed/agent),
//             every storing wave waits s_waitcnt vmcnt(0), the work-group's barrier, then ONE lane stores the counter sc1;
//   consumer: one lane polls the counter with sc1 loads, the work-group's barrier, then EVERY load of the handed-off
//             bytes is an sc1 load -- to registers (__hip_atomic_load relaxed/agent on a global pointer) or, since the
//             end of round 3, straight into LDS (global_load_lds_dwordx4 ... sc1 in chunk_asm.h: the same cache policy
//             bits on the same load path; the wave waits vmcnt(0) and the work-group's barrier before any lane reads
//             the LDS image).
// No agent-scope release (buffer_wbl2: it would write back every dirty line of the XCD's L2, the other work-groups'
// private results included) and no acquire (buffer_inv: it empties the XCD's L2 for all 16 work-groups sharing it,
// twice per step each) -- measured in round 2: 216 -> ... us per 512-column sub-panel.
//
// "One work-group per CU" in the guide's table is the regime the form was MEASURED in.  panel_ll_kernel meets it for
// its own work-groups (148 KB of LDS each); foreign work-groups that fit beside one (no LDS to speak of, <= 152 registers) do
// not change the argument, because it never relied on the CU's L1: every load of a handed-off byte is an sc1 load (served
// past the L1) and every such byte was stored sc1 (written through, dropped from the producer's L2).
// The fenced form (-DGPIRT_PANEL_FENCES, `make fences`) is built and compared bit for bit by tests/test_gpu_fences.py.
//
// Rules that keep such kernels safe on this hardware:
//   * a work-group only ever waits for work-groups with a SMALLER block index (dispatched earlier), and the
//     awaited result is always produced before its producer waits for anything itself;
//   * every spin is bounded: after SPIN_LIMIT polls the waiter raises info[1] and the whole work-group
//     leaves, so the grid always drains and the host reports the failure instead of hanging the GPU.
#pragma once

#include "common.h"

namespace gpirt {

constexpr int SPIN_LIMIT = 1 << 22;        // polls (each with an s_sleep) before giving up: ~ seconds

__device__ __forceinline__ unsigned long long ld_prog(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All 256 threads call it.  Returns false when the wait expired (uniform across the work-group).
// `have` caches the last value seen for this counter so later waits on smaller values cost nothing.
// `urgent` (uniform): the waiter is the next diagonal owner -- it polls back to back; everybody else naps
// between polls, which also keeps them from crowding the memory channel that holds the fresh block.
// On expiry the first work-group to give up records who waited for what in info[2..7] (api.hip:
// report_panel_guard): waiting row block, awaited row block, value needed, value last seen.
__device__ __forceinline__ bool wait_prog(const unsigned long long* p, unsigned long long need,
                                          unsigned long long& have, unsigned long long* s_seen, int* info,
                                          bool urgent = false, int waiter = -1, int awaited = -1)
{
    if (have >= need) return true;
    if (threadIdx.x == 0) {
        unsigned long long v = ld_prog(p);
        int spins = 0;
        while (v < need && ++spins < SPIN_LIMIT) {
            if (urgent) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(24);
            v = ld_prog(p);
        }
        if (v < need) {
            if (atomicCAS(info + 1, 0, 1) == 0) {
                info[2] = waiter; info[3] = awaited;
                info[4] = (int)(need & 0xffffffffu); info[5] = (int)(need >> 32);
                info[6] = (int)(v & 0xffffffffu);    info[7] = (int)(v >> 32);
            }
            v = 0;
        }
        *s_seen = v;
    }
    __syncthreads();
    const unsigned long long v = *s_seen;
#ifdef GPIRT_PANEL_FENCES
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale L1/L2 lines before reading the data
#endif
    __syncthreads();                                  // s_seen may be rewritten by the next wait
    if (v < need) return false;
    have = v;
    return true;
}

// ONE thread polls (back to back) until the counter reaches `need`; returns the value seen, 0 when the wait expired
// (the guard is recorded as in wait_prog).  The caller hands the result to the work-group through LDS + a barrier.
__device__ __forceinline__ unsigned long long poll_prog(const unsigned long long* p, unsigned long long need, int* info,
                                                        int waiter, int awaited)
{
    unsigned long long v = ld_prog(p);
    int spins = 0;
    while (v < need && ++spins < SPIN_LIMIT) {
        __builtin_amdgcn_s_sleep(1);
        v = ld_prog(p);
    }
    if (v < need) {
        if (atomicCAS(info + 1, 0, 1) == 0) {
            info[2] = waiter; info[3] = awaited;
            info[4] = (int)(need & 0xffffffffu); info[5] = (int)(need >> 32);
            info[6] = (int)(v & 0xffffffffu);    info[7] = (int)(v >> 32);
        }
        v = 0;
    }
    return v;
}

// every wave's (sc1) stores have been accepted, then one lane raises the row block's counter
__device__ __forceinline__ void publish(unsigned long long* p, unsigned long long value)
{
#ifdef GPIRT_PANEL_FENCES
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(p, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

}  // namespace gpirt
