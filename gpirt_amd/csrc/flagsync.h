// flagsync.h -- progress counters between the work-groups of ONE persistent kernel (panel.hip): release /
// acquire at agent scope through device memory.
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
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale L1/L2 lines before reading the data
    __syncthreads();                                  // s_seen may be rewritten by the next wait
    if (v < need) return false;
    have = v;
    return true;
}

// make this work-group's global stores visible, then raise the row block's counter
__device__ __forceinline__ void publish(unsigned long long* p, unsigned long long value)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(p, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace gpirt
