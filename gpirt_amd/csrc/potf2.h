// potf2.h -- 64 x 64 Cholesky of a diagonal block by one 256-thread work-group (device function, so the
// panel-update GEMM can run it as the epilogue of the tile that produces the next diagonal block).
#pragma once

#include "common.h"

namespace gpirt {

// ------------------------------------------------------------------ diagonal block ---------
// 64 x 64 Cholesky in one work-group, register tiled: thread (tr, tc) = (t & 15, t >> 4) owns the
// 4 x 4 tile rows 4tr.., columns 4tc.. .  Per 4-column panel jb: the 16 lanes that own it (one
// contiguous 16-lane group of one wavefront) factor it with v_readlane broadcasts of the pivot
// row -- no barrier inside the panel --, publish it to LDS, and after ONE barrier every trailing
// tile applies the rank-4 update from registers.  16 barriers in total (the unblocked version
// this replaces needed 192 and ran 83 us; see profiles/).
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// sPm: 2 * 4 * 64 doubles of LDS (16-byte aligned), sfail_p: one int of LDS.  All 256 threads of the
// work-group must call it.
__device__ __forceinline__ void potf2_64_body(double* __restrict__ A, int64_t lda, int nb, int k0,
                                              int* __restrict__ info, double* __restrict__ sPm,
                                              int* __restrict__ sfail_p)
{
    constexpr int NBI = 64;
    double (*sP)[4 * NBI] = reinterpret_cast<double (*)[4 * NBI]>(sPm);   // sP[buf][k * 64 + row]
    int& sfail = *sfail_p;
    __builtin_amdgcn_s_setprio(3);   // latency-bound chain: win issue arbitration against co-resident GEMM waves
    const int t = threadIdx.x;
    const int tr = t & 15, tc = t >> 4;
    double a[4][4];                                                   // a[i][k]: row 4tr+i, col 4tc+k
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * tr + i, col = 4 * tc + k;
            double v = (row == col) ? 1.0 : 0.0;                      // identity padding for nb < 64
            if (row >= col && row < nb && col < nb) v = A[(int64_t)row + (int64_t)col * lda];
            a[i][k] = v;
        }
    if (t == 0) sfail = 0x7fffffff;
    __syncthreads();
    int fail = 0x7fffffff;
#pragma unroll
    for (int jb = 0; jb < 16; ++jb) {
        const int buf = jb & 1;
        if (tc == jb && tr >= jb) {
            // The 16-lane group of panel jb.  Lane `src` (tr == jb) holds the 4 x 4 diagonal tile and
            // factors it in place with static indices; the lanes below (tr > jb) hold full tiles and
            // need no per-element predicates -- the two roles diverge once per column.
            const int src = jb + 16 * (jb & 3);
            const bool is_diag = (tr == jb);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * jb + jj;
                const double d = readlane_f64(a[jj][jj], src);
                if (!(d > 0.0) && j < nb && fail == 0x7fffffff) fail = j + 1;
                // pivot column scaled by 1/sqrt(d) (LAPACK dpotf2 scales by the reciprocal too);
                // rsqrt keeps the 64-step dependent chain short.  d <= 0 gives NaN: propagates.
                const double rinv = (d > 0.0) ? rsqrt(d) : __builtin_nan("");
                if (is_diag) {
                    a[jj][jj] = d * rinv;
#pragma unroll
                    for (int i = jj + 1; i < 4; ++i) a[i][jj] *= rinv;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i][jj] *= rinv;
                }
#pragma unroll
                for (int jj2 = jj + 1; jj2 < 4; ++jj2) {
                    const double lc = readlane_f64(a[jj2][jj], src);  // L[4jb+jj2][j]
                    if (is_diag) {
#pragma unroll
                        for (int i = jj2; i < 4; ++i) a[i][jj2] = fma(-a[i][jj], lc, a[i][jj2]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[i][jj2] = fma(-a[i][jj], lc, a[i][jj2]);
                    }
                }
            }
            // publish the panel: sP[buf][k * 64 + row]; the strict upper part of the diagonal tile is zero
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double4 v;
                v.x = (is_diag && 0 < k) ? 0.0 : a[0][k];
                v.y = (is_diag && 1 < k) ? 0.0 : a[1][k];
                v.z = (is_diag && 2 < k) ? 0.0 : a[2][k];
                v.w = a[3][k];
                *reinterpret_cast<double4*>(&sP[buf][k * NBI + 4 * tr]) = v;
            }
        }
        __syncthreads();
        if (tc > jb && tr >= tc) {
            double4 lr[4], lcn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lr[k] = *reinterpret_cast<const double4*>(&sP[buf][k * NBI + 4 * tr]);
                lcn[k] = *reinterpret_cast<const double4*>(&sP[buf][k * NBI + 4 * tc]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double r4[4] = { lr[k].x, lr[k].y, lr[k].z, lr[k].w };
                const double c4[4] = { lcn[k].x, lcn[k].y, lcn[k].z, lcn[k].w };
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) a[i][kk] = fma(-r4[i], c4[kk], a[i][kk]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * tr + i, col = 4 * tc + k;
            if (row >= col && row < nb && col < nb) A[(int64_t)row + (int64_t)col * lda] = a[i][k];
        }
    if (fail != 0x7fffffff) atomicMin(&sfail, fail);
    __syncthreads();
    if (t == 0 && sfail != 0x7fffffff) atomicCAS(info, 0, k0 + sfail);
}


}  // namespace gpirt

namespace gpirt {

// ------------------------------------------------------------------ diagonal block, LDS resident ----
// potf2_64_lds: the same 64 x 64 Cholesky, for a block that already sits in LDS (column-major, stride LS,
// identity-padded beyond nb) -- the persistent panel kernel (panel.hip) hands its register accumulators
// over this way instead of through global memory.  Organised for latency, 16 columns at a time:
//   * wave 0 holds block column b with ONE ROW PER LANE (16 registers) and runs the 16 pivot steps without a
//     barrier; two v_readlane pairs per column sit on the chain, the other multipliers are LDS broadcasts;
//   * the rank-16 update of the remaining 16 x 16 blocks is fp64 MFMA (one block = 4 instructions):
//     first the blocks of column b + 1 (the only ones the next pivot steps need), then -- while wave 0
//     already factors column b + 1 -- the rest, on waves 1..3.
// 8 barriers in total.  L's lower triangle is stored to Aout (global) straight from wave 0's registers and
// left in sD.  A non-positive pivot records k0 + (1-based column) in *info (first failure wins) and NaNs
// propagate, like LAPACK dpotf2 + the old kernel.  All 256 threads must call it.  late_flag / late_value:
// optional progress counter of the caller to raise once every wave's EARLIER global stores are visible.
// col_flag / col_base: optional second counter, raised to col_base + b + 1 as soon as block column b (b = 0, 1, 2)
// of L is in memory -- the next diagonal owner solves against L column block by column block while the later
// pivots are still running (panel.hip).  Such a column is stored by wave 3 ALONE, which waits for its own stores
// (one block column later) and raises the counter without a barrier.  winv (4 x 256 doubles of global memory) / sW (the same in LDS):
// the inverses W_b of the four 16 x 16 diagonal blocks, dense column-major, built by wave 1 in step with the pivots and
// stored in front of the counter that announces their column -- the consumer applies them with MFMAs and never inverts
// anything itself (COLS = true; all three pointers are then required).  The caller zeroes the hand-shake slots (sXT[p * 18 + 16], p = 0..63) before its barrier in front of the call.
template <int LS>
__device__ __forceinline__ void potf2_lds_update_block(double* __restrict__ sD, int b, int R, int C)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const int pi = 4 * (i & 3) + (i >> 2);
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = sD[(16 * C + 4 * g + r) * LS + 16 * R + i];
    double a[4], bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a[s] = -sD[(16 * b + 4 * g + s) * LS + 16 * C + pi];
        bv[s] = sD[(16 * b + 4 * g + s) * LS + 16 * R + i];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bv[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) sD[(16 * C + 4 * g + r) * LS + 16 * R + i] = acc[r];
}

// block column b of the factored block (rows >= column) from LDS to global memory, one row per lane: 16 sc1 stores
template <int LS>
__device__ __forceinline__ void potf2_store_column(const double* __restrict__ sD, double* Aout, int64_t lda, int nb, int b)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int c = 16 * b + cc;
        if (lane >= c && lane < nb && c < nb)
            __hip_atomic_store(&Aout[lane + (int64_t)c * lda], sD[c * LS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// 1/sqrt(d) for the pivot chain: hardware estimate + one third-order correction (the same arithmetic as the
// library rsqrt, minus its special-case select): d <= 0, NaN and Inf all come out as NaN/Inf and propagate.
__device__ __forceinline__ double rsqrt_chain(double d)
{
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

constexpr int POTF2_XS = 18;      // row stride of the multiplier copy sXT (64 x 18 doubles of LDS)
template <int LS, bool COLS = false>
__device__ __forceinline__ void potf2_64_lds(double* __restrict__ sD, double* __restrict__ sXT, double* Aout,
                                             int64_t lda, int nb, int k0, int* __restrict__ info,
                                             unsigned long long* late_flag = nullptr, unsigned long long late_value = 0,
                                             long long* dbg = nullptr,
                                             unsigned long long* col_flag = nullptr, unsigned long long col_base = 0,
                                             double* winv = nullptr, double* __restrict__ sW = nullptr)
{
    constexpr int XS = POTF2_XS;
#define POTF2_STAMP(slot) do { if (dbg && threadIdx.x == 0) dbg[4 * b + (slot)] = wall_clock64(); } while (0)
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    int fail = 0;
    double mine = 0.0;                 // (COLS) 1 / L_pp of this lane's own pivot, once it is known
    // One block column.  COLS = false: unrolled four times (block index a constant everywhere).  COLS = true: a real
    // loop -- with all four waves busy in different code, four copies of it no longer fit the instruction cache
    // (measured inside the panel kernel: 2.0-2.3 us of pivots per block column unrolled, 1.6 rolled; 1.3-1.6 for the
    // COLS = false kernel, whose waves 1..3 mostly sit at barriers).
    auto block_column = [&](const int b) {
        if (wave == 0) {
            __builtin_amdgcn_s_setprio(3);
            POTF2_STAMP(0);
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = sD[(16 * b + c) * LS + lane];
            // Left-looking inside the block column, software-pipelined so that nothing but the pivot itself
            // (readlane -> fma -> readlane -> rsqrt -> scale) is ever waited for:
            //   * column c+1 collects the updates of columns 0 .. c-1 while pivot c is in flight; the multipliers
            //     L[p+1][k] come from sXT (row-major copy of the finished columns; a uniform-address LDS read is
            //     a broadcast, no SGPR traffic), k <= c-2 fetched a whole column earlier, k = c-1 by v_readlane;
            //   * the reads for column c+2 are issued now and consumed in the next iteration.
            double pre = x[0];
            double mA[16], mB[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int p = 16 * b + c;
                if (c + 2 < 16) {                      // (a) multipliers of column c+2, k = 0 .. c-1
#pragma unroll
                    for (int k = 0; k < c; k += 2) {
                        const double2 mm = *reinterpret_cast<const double2*>(&sXT[(p + 2) * XS + k]);
                        mB[k] = mm.x;
                        mB[k + 1] = mm.y;
                    }
                }
                double acc = pre;                      // (b) the chain
                if (c >= 1) acc = fma(-x[c - 1], readlane_f64(x[c - 1], p), acc);
                const double d = readlane_f64(acc, p);
                if (!(d > 0.0) && p < nb && fail == 0) fail = p + 1;
                const double rinv = rsqrt_chain(d);
                if (c + 1 < 16) {                      // (c) column c+1 minus columns 0 .. c-1
                    double s0 = x[c + 1], s1 = 0.0;
#pragma unroll
                    for (int k = 0; k + 1 < c; k += 2) {
                        s0 = fma(-x[k], mA[k], s0);
                        if (k + 2 < c) s1 = fma(-x[k + 1], mA[k + 1], s1);
                    }
                    if (c >= 1) s1 = fma(-x[c - 1], readlane_f64(x[c - 1], p + 1), s1);
                    pre = s0 + s1;
                }
                x[c] = acc * rinv;                     // (d)
                if (COLS) {
                    // hand column p to the inverse wave: the column, then 1/L_pp into row p's pad slot 16, which the
                    // inverse wave polls (zero = not yet; the caller zeroes the slots).  Every lane rewrites ITS OWN
                    // slot with `mine` (its pivot's reciprocal once it has happened, zero before): no branch and no
                    // pivot-dependent address on the chain.  The LDS executes one wave's accesses in order and the
                    // empty asm keeps the compiler from reordering them (volatile accesses would each be waited for:
                    // measured 36 us per block instead of 10).
                    sXT[lane * XS + c] = x[c];
                    asm volatile("" ::: "memory");
                    mine = (lane == p) ? rinv : mine;
                    sXT[lane * XS + 16] = mine;
                    asm volatile("" ::: "memory");
                } else {
                    sXT[lane * XS + c] = x[c];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) mA[k] = mB[k];
            }
            POTF2_STAMP(1);
#pragma unroll
            for (int c = 0; c < 16; ++c) sD[(16 * b + c) * LS + lane] = x[c];
            POTF2_STAMP(2);
            __builtin_amdgcn_s_setprio(0);
            // deferred publication (see below): every wave makes sure its earlier global stores have been
            // accepted by the L2 (long done by now) before the first barrier
            if (b == 0 && late_flag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (b == 0 && late_flag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (COLS) {
                // rest of the previous block column's update (columns b + 1 ..) on wave 2, hidden behind wave 0's pivots
                if (wave == 2) {
                    if (b == 1) { potf2_lds_update_block<LS>(sD, 0, 2, 2); potf2_lds_update_block<LS>(sD, 0, 3, 3); potf2_lds_update_block<LS>(sD, 0, 3, 2); }
                    if (b == 2) potf2_lds_update_block<LS>(sD, 1, 3, 3);
                }
                // W_b = L_bb^-1 on wave 1, one column of L behind wave 0's pivots: lane j runs the forward substitution on
                // e_j, step k as soon as pivot 16b + k is out (LDS counter), so W_b is complete ~one step after the last pivot
                if (wave == 1 && lane < 16) {
                    double w[16];
#pragma unroll
                    for (int i2 = 0; i2 < 16; ++i2) w[i2] = (i2 == lane) ? 1.0 : 0.0;
                    int zero;                                        // opaque: keeps the addresses below as
                    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));     // (one VGPR base) + immediate offsets
                    const double* blk = sXT + (16 * b) * XS + zero;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        double rk;
                        while (true) {
                            asm volatile("" ::: "memory");
                            rk = blk[k * XS + 16];
                            if (rk != 0.0) break;                  // (NaN from a failed pivot passes too)
                            __builtin_amdgcn_s_sleep(1);
                        }
                        asm volatile("" ::: "memory");
                        double lk[16];
#pragma unroll
                        for (int i2 = k + 1; i2 < 16; ++i2) lk[i2] = blk[i2 * XS + k];
                        const double wk = w[k] * rk;
                        sW[256 * b + lane * 16 + k] = wk;          // (final: written as it appears, nothing is left for the end)
#pragma unroll
                        for (int i2 = k + 1; i2 < 16; ++i2) w[i2] = fma(-lk[i2], wk, w[i2]);
                    }
                }
            } else if (b == 1) {                           // (wave 3 is busy publishing, see below)
                if (wave == 1) { potf2_lds_update_block<LS>(sD, 0, 2, 2); potf2_lds_update_block<LS>(sD, 0, 3, 3); }
                if (wave == 2) potf2_lds_update_block<LS>(sD, 0, 3, 2);
            } else if (b == 2) {
                if (wave == 1) potf2_lds_update_block<LS>(sD, 1, 3, 3);
            }
        }
        __syncthreads();
        if (b < 3) {
            // blocks (R, b + 1), R = b + 1 .. 3: one per wave, waves 1..3
            const int R = b + wave;
            if (wave >= 1 && R <= 3) potf2_lds_update_block<LS>(sD, b, R, b + 1);
            __syncthreads();
        }
        // L's block column b goes out to global memory from LDS on waves 1..3 (write-through: other
        // work-groups read it next), off wave 0's chain except for the last block column
        // The caller's last result block (stored to global before the call) is published from here instead of
        // before the call: all waves' stores are in the L2 (vmcnt(0) before the first barrier), wave 3 -- idle
        // for the next block column -- writes the L2 back (release fence, ~2.5 us) and raises the counter,
        // while wave 0 is already deep in the next pivots.
        if (b == 0 && late_flag && wave == 3) {
#ifdef GPIRT_PANEL_FENCES
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
            // (fence-free form, flagsync.h: every wave waited vmcnt(0) on its sc1 stores before the barrier above)
            if (lane == 0) __hip_atomic_store(late_flag, late_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (COLS && b < 3) {
            // progressive hand-off: column b and W_b go out on wave 3 alone, which waits for its own stores and raises
            // the column counter without a barrier
            if (wave == 3) {
                // (the counter of column b - 1 first: its stores went out a whole block column ago, so the wait is
                // over before it starts -- waiting right behind the stores held wave 3, and with it the next barrier
                // of the pivot wave, for the 2-3 us a write-through takes while the panel's other work-groups stream)
                if (b >= 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(col_flag, col_base + (unsigned long long)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                potf2_store_column<LS>(sD, Aout, lda, nb, b);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __hip_atomic_store(&winv[256 * b + lane + 64 * q], sW[256 * b + lane + 64 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (COLS) {
            // last block column: W_3 on wave 1, the column itself on waves 2 and 3 (the caller publishes)
            if (wave == 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(col_flag, col_base + 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (wave == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __hip_atomic_store(&winv[256 * 3 + lane + 64 * q], sW[256 * 3 + lane + 64 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (wave >= 2) {
                // one row per lane, eight columns per wave: the column index is wave-uniform, so the address is
                // (scalar column offset) + lane -- a per-lane column index costs a 64-bit multiply per store, which
                // the compiler hoists to the top of the kernel and spills
                const int c0 = 48 + 8 * (__builtin_amdgcn_readfirstlane(wave) - 2);
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    const int c = c0 + cc;
                    if (lane >= c && lane < nb && c < nb)
                        __hip_atomic_store(&Aout[lane + (int64_t)c * lda], sD[c * LS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else if (wave >= 1) {
            const int u = (wave - 1) * 64 + lane;          // 0 .. 191
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int idx = u + 192 * q;               // 16 columns x 64 rows = 1024 elements
                const int r = idx & 63, c = 16 * b + (idx >> 6);
                if (idx < 1024 && r >= c && r < nb && c < nb)
                    __hip_atomic_store(&Aout[r + (int64_t)c * lda], sD[c * LS + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        POTF2_STAMP(3);
    };
    if (COLS) {
#pragma unroll 1
        for (int b = 0; b < 4; ++b) block_column(b);
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) block_column(b);
    }
#undef POTF2_STAMP
    if (wave == 0 && lane == 0 && fail != 0) atomicCAS(info, 0, k0 + fail);
}

}  // namespace gpirt
