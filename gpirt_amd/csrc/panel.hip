// panel.hip -- one outer panel of the blocked Cholesky (arma::chol(S,"lower"), src/gpirtMCMC.cpp:17,78,97)
// factored by ONE persistent kernel instead of a chain of ~3 launches per 64 columns.
//
// The panel is columns [K0, c1) (<= 1024 of them) and every row from K0 down.  It is cut into 64-row
// blocks; work-group w owns row block w (w + G, w + 2G, ... when there are more blocks than resident
// work-groups) and sweeps it LEFT-LOOKING over the panel's 64-column blocks j = 0, 1, ...:
//
//     T      = A[R, j] - sum_{k<j} X[R, k] X[j, k]^T        (fp64 MFMA, accumulators in registers)
//     X[R,j] = T L_jj^{-T}                                   (substitution in the MFMA layout, solve64.h)
//
// and the owner of the diagonal row block j finishes with  D = A[j,j] - sum_k X[j,k] X[j,k]^T (kept in
// registers, updated as each X[j,k] appears) and the 64 x 64 Cholesky L_jj = chol(D) (potf2.h).
// Row block R only ever needs finished pieces of the diagonal owners j < R: X[j, k] and L_jj.  Those are
// handed over through memory with one progress counter per row block (release / acquire at agent scope;
// measured hand-off ~1 us across XCDs): no grid-wide barrier, no launch on the critical path.  The
// pivot chain  L_jj -> X[j+1,j] -> D_{j+1} -> L_{j+1,j+1}  therefore runs inside two work-groups while
// every other work-group streams its GEMMs behind it.
//
// Deadlock freedom: dependencies point to strictly smaller row-block indices, the diagonal owners are the
// first <= 16 work-groups of the grid (dispatched first), and the grid never exceeds the number of CUs.
// Every spin is bounded (GPIRT_SPIN_LIMIT polls): on expiry the work-group raises info[1] and leaves, so
// the grid always drains; the host turns info[1] into an error.  A non-positive pivot is recorded in
// info[0] like before and NaNs propagate -- the counters are still published, nobody waits forever.
#include "common.h"
#include "kernels.h"
#include "solve64.h"
#include "potf2.h"
#include "flagsync.h"
#include "chunk_asm.h"

namespace gpirt {

namespace {

constexpr int PB = 64;                     // block size

struct PanelArgs {
    double* A; int64_t lda; int64_t n;
    int64_t K0, c1;                        // panel columns [K0, c1)
    unsigned long long* prog;              // one counter per absolute 64-row block
    unsigned long long* qprog;             // a second one: finished 16-column blocks of the row block's L_RR (base + 1..3)
    double* winv;                          // [ABSOLUTE column block][4][256]: inverses of the 16 x 16 diagonal blocks of L_jj
                                           // (one slot per 64-column block of the matrix: a rows kernel that runs late still
                                           // finds its sub-panel's inverses after later sub-panels have been factored)
    unsigned long long base;               // epoch of this launch: counters below base are stale
    int* info;
    int nrb;                               // row blocks of this panel (rows K0 .. n-1)
    int ncb;                               // column blocks of this panel
    int rb_begin, rb_end;                  // this launch sweeps the panel's row blocks [rb_begin, rb_end) (relative to K0)
    long long* trace;                      // optional (tools/micro/panel_bench.hip): [row block][step][8] 100 MHz stamps
};

#define PANEL_STAMP(slot)                                                                         \
    do {                                                                                          \
        if (p.trace && threadIdx.x == 0) p.trace[((int64_t)Rr * 40 + (tstep)) * 8 + (slot)] = wall_clock64(); \
    } while (0)

// The thread index as the helpers below see it: re-materialised at every call (the empty asm cannot be hoisted or merged),
// so the per-lane offsets derived from it are recomputed where they are used -- a few VALU operations -- instead of being
// hoisted to the top of the kernel, where dozens of them stay alive for its whole length and push the accumulators of the
// hot loops out to scratch (the kernel runs at 512 registers per lane).
__device__ __forceinline__ int tid_here()
{
    int v = (int)threadIdx.x;
    asm volatile("" : "+v"(v));
    return v;
}

// Global accesses are written as (wave-uniform base) + (32-bit per-lane byte offset): the column part of
// every address is scalar arithmetic, the lane offset is computed once, and out-of-range rows are clamped
// to the last row and zeroed after the load -- no predicated loads, so a block's 16 loads fly together.
__device__ __forceinline__ double ldg_off(const double* sbase, uint32_t voff)
{
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(sbase) + voff);
}
// ... and what ANOTHER work-group stored (its X blocks, L_jj) is read with sc1 loads: straight from memory, never from a
// line this XCD's L2 may have kept from before the store (flagsync.h)
__device__ __forceinline__ double ldg_sc1(const double* sbase, uint32_t voff)
{
#ifdef GPIRT_PANEL_FENCES
    return ldg_off(sbase, voff);
#else
    return __hip_atomic_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(sbase) + voff), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// OWN ROWS ONLY: a plain store stays in this XCD's L2 until the kernel ends, so the bytes may be re-read by the storing
// work-group alone (its X[R, k] operands of later steps) -- never by another work-group of this launch or of a launch
// running beside it; whatever a neighbour reads goes through stg_off.  (tests/test_panel_isa.py counts the plain and the
// sc1 accesses of every kernel in this file against a committed census, so a new plain access does not slip in unseen.)
__device__ __forceinline__ void stg_plain(double* sbase, uint32_t voff, double x)
{
    *reinterpret_cast<double*>(reinterpret_cast<char*>(sbase) + voff) = x;
}
// Results other work-groups will read are stored write-through at agent scope (sc1): nothing of ours stays
// dirty in the XCD's L2, so the release fence before a counter update has (almost) nothing to write back.
__device__ __forceinline__ void stg_off(double* sbase, uint32_t voff, double x)
{
    __hip_atomic_store(reinterpret_cast<double*>(reinterpret_cast<char*>(sbase) + voff), x, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// 64 x 64 block of A at (row0, col0), all 64 columns valid -> 16 doubles per thread, ready for store_block_lds
__device__ __forceinline__ void load_block_regs(double (&v)[16], const double* A, int64_t lda,
                                                int64_t n, int64_t row0, int64_t col0)
{
    const int t = tid_here();
    const int r = t & 63;
    const int last = (int)(n - 1 - row0);
    const bool live = r <= last;
    const uint32_t voff = (uint32_t)(((live ? r : last) + (int64_t)(t >> 6) * lda) * 8);
    if (row0 + PB <= n) {        // full block (uniform): plain loads, nothing between them and their first use
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = ldg_sc1(A + row0 + (col0 + 4 * q) * lda, voff);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double x = ldg_sc1(A + row0 + (col0 + 4 * q) * lda, voff);
            v[q] = live ? x : 0.0;
        }
    }
}
__device__ __forceinline__ void store_block_lds(const double (&v)[16], double* __restrict__ sT)
{
    const int t = tid_here();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int idx = t + 256 * q;
        sT[(idx >> 6) * S64_LS + (idx & 63)] = v[q];
    }
}

// the wave's 16-row strip of a 64-column block, in the accumulator layout of solve64.h:
// lane (i, g) gets columns 16J + 4g + r of row  row0 + 16 wave + i.  ncols < 64 only for the matrix's
// last, ragged block (then the column index is clamped per lane).
__device__ __forceinline__ void load_strip(d4 (&X)[4], const double* A, int64_t lda, int64_t n,
                                           int64_t row0, int64_t col0, int ncols)
{
    const int t_ = tid_here(), lane = t_ & 63, wave = t_ >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int rr = 16 * wave + i;
    const int last = (int)(n - 1 - row0);
    const bool live = rr <= last;
    const int rc = live ? rr : last;
    if (ncols == PB && row0 + PB <= n) {
        const uint32_t voff = (uint32_t)((rr + (int64_t)(4 * g) * lda) * 8);
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[J][r] = ldg_off(A + row0 + (col0 + 16 * J + r) * lda, voff);
    } else if (ncols == PB) {
        const uint32_t voff = (uint32_t)((rc + (int64_t)(4 * g) * lda) * 8);
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double x = ldg_off(A + row0 + (col0 + 16 * J + r) * lda, voff);
                X[J][r] = live ? x : 0.0;
            }
    } else {
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * J + 4 * g + r;
                const bool ok = live && c < ncols;
                const double x = ldg_off(A + row0 + rc + (col0 + (c < ncols ? c : 0)) * lda, 0u);
                X[J][r] = ok ? x : 0.0;
            }
    }
}
// shared: other work-groups will read the strip (a diagonal owner's X) -> sc1 write-through; a row block nobody else
// reads inside this kernel stores plainly, so its own re-reads (XI) hit the L2
__device__ __forceinline__ void store_strip(const d4 (&X)[4], double* A, int64_t lda, int64_t n,
                                            int64_t row0, int64_t col0, int ncols, bool lower_only, bool shared = true)
{
    const int t_ = tid_here(), lane = t_ & 63, wave = t_ >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int rr = 16 * wave + i;
    if (row0 + rr >= n) return;
    const uint32_t voff = (uint32_t)((rr + (int64_t)(4 * g) * lda) * 8);
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * J + 4 * g + r;
            if (c < ncols && (!lower_only || rr >= c)) {
#ifdef GPIRT_PANEL_FENCES
                stg_off(A + row0 + (col0 + 16 * J + r) * lda, voff, X[J][r]);
#else
                if (shared) stg_off(A + row0 + (col0 + 16 * J + r) * lda, voff, X[J][r]);
                else stg_plain(A + row0 + (col0 + 16 * J + r) * lda, voff, X[J][r]);
#endif
            }
        }
}

// ---- progressive hand-off of L_jj to the NEXT diagonal owner (the pivot chain) ---------------------------------
// 16-column block b of L_jj (full 64-row block), 4 doubles per thread, sc1 loads / the same staged into LDS
__device__ __forceinline__ void load_lcol(double (&v)[4], const double* A, int64_t lda, int64_t orow0, int64_t col0, const int b)
{
    const int t = tid_here();
    const uint32_t voff = (uint32_t)(((t & 63) + (int64_t)(t >> 6) * lda) * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ldg_sc1(A + orow0 + (col0 + 16 * b + 4 * q) * lda, voff);
}
__device__ __forceinline__ void stage_lcol(const double (&v)[4], double* sM, const int b)
{
    const int t = tid_here(), r = t & 63, cq = t >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 16 * b + cq + 4 * q;
        if (r >= 16 * b + 16) sM[c * S64_LS + r] = v[q];       // the diagonal block arrives as its inverse
    }
}

// the wave's 16-row strip of a 64 x 64 block kept column-major in LDS (accumulator layout <-> sB[c * S64_LS + row])
__device__ __forceinline__ void strip_from_lds(d4 (&X)[4], const double* sB)
{
    const int t_ = tid_here(), lane = t_ & 63, wave = t_ >> 6;
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) X[J][r] = sB[(16 * J + 4 * g + r) * S64_LS + 16 * wave + i];
}
__device__ __forceinline__ void strip_to_lds(const d4 (&X)[4], double* sB)
{
    const int t_ = tid_here(), lane = t_ & 63, wave = t_ >> 6;
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) sB[(16 * J + 4 * g + r) * S64_LS + 16 * wave + i] = X[J][r];
}

__global__ __launch_bounds__(256) void panel_ll_kernel(PanelArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // Two 64 x 64 buffers (34 KB each): the "other" GEMM operand alternates between them, and whichever
    // one the last chunk did not use then takes L_jj for the solve and X for the diagonal update; a third holds D
    // and a fourth stages the chunks' own-row operand (below).  148 KB in all (114 without the fourth), so two of
    // these work-groups never share a CU (160 KB).  The kernel runs at 329 registers per lane (<true>; 422 <false>), not
    // the full 512: a small foreign work-group (<= 12 KB of LDS, <= 176 registers -- a split-K sum, a copy-back, a flag
    // store) CAN sit beside it.  That does not touch the hand-off protocol: no handed-off
    // byte is ever read through the CU's L1 or expected in this XCD's L2 (every such store and load is sc1, flagsync.h), so
    // what a co-resident work-group may have left in L1 cannot be observed; exclusivity only matters for speed.
    double* sT0 = smem;
    double* sT1 = smem + PB * S64_LS;
    double* sXT = smem + 2 * PB * S64_LS;             // potf2's multiplier copy
    // The diagonal block D of a diagonal owner lives HERE between its steps, not in 32 accumulator registers:
    // each step reads its strip, updates it and writes it back (own rows only: no barrier), potf2 factors it in place,
    // and the register allocator no longer spills D around the chunk loop (its reload sat on the pivot chain)
    double* sDD = smem + 2 * PB * S64_LS + PB * POTF2_XS;
    // Staging area of the chunks' own-row operand: one slice of 8 x 136 doubles per wave (below)
    double* sXI = smem + 3 * PB * S64_LS + PB * POTF2_XS;
    __shared__ unsigned long long s_seen;

    const int t = tid_here();
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 15, g = lane >> 4;
    double* A = p.A;      // read and written by many work-groups: no restrict anywhere in this file
    const int64_t lda = p.lda, n = p.n;
    const int cb0 = (int)(p.K0 / PB);                 // absolute index of the panel's first block
    const int n_all = p.rb_end - p.rb_begin;
    for (int idx = blockIdx.x; idx < n_all; idx += gridDim.x) {
        const int Rr = p.rb_begin + idx;
        const int64_t row0 = p.K0 + (int64_t)Rr * PB;
        const bool is_diag = Rr < p.ncb;
        const int jend = is_diag ? Rr : p.ncb;        // off-diagonal column blocks of this row block
        unsigned long long* my_prog = p.prog + cb0 + Rr;
        d4 D[4];
        int dcols = 0;
        if (is_diag) {
            const int64_t left = p.c1 - row0;
            dcols = (int)(left < PB ? left : PB);
            load_strip(D, A, lda, n, row0, row0, dcols);
#pragma unroll
            for (int J = 0; J < 4; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * J + 4 * g + r, rr = 16 * wave + i;
                    if (rr >= dcols || c >= dcols) D[J][r] = (rr == c) ? 1.0 : 0.0;      // identity past a ragged end
                }
            strip_to_lds(D, sDD);
        }
        for (int j = 0; j < jend; ++j) {
            const int t = tid_here();                 // (shadows the kernel's: per-lane offsets are rebuilt per step)
            const int lane = t & 63, wave = t >> 6;
            const int i = lane & 15, g = lane >> 4;
            const int64_t col0 = p.K0 + (int64_t)j * PB;
            const int64_t orow0 = col0;               // the diagonal owner of column block j sits at rows col0..
            const unsigned long long* o_prog = p.prog + cb0 + j;
            unsigned long long have = 0;
            double* sM = (j & 1) ? sT1 : sT0;         // free: the last chunk (k = j - 1) reads the other buffer
            const int tstep = j;
            PANEL_STAMP(0);
            const int jcols = (int)((p.c1 - col0) < PB ? (p.c1 - col0) : PB);   // < 64 only in the matrix's last block
            d4 T[4];
            load_strip(T, A, lda, n, row0, col0, jcols);
            // ---- T -= sum_k X[R,k] X[j,k]^T, chunks of 64, the other operand double-buffered through LDS
            if (j > 0) {
                if (!wait_prog(o_prog, p.base + 1, have, &s_seen, p.info, false, cb0 + Rr, cb0 + j)) return;
                const bool full_blocks = (row0 + PB <= n) && (orow0 + PB <= n);
#ifdef PANEL_CHUNK_PROF
                long long cp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#define CP_MARK_NOWAIT(slot) do { if (p.trace && threadIdx.x == 0) { const long long now_ = wall_clock64(); cp[slot] += now_ - cp_last; cp_last = now_; } } while (0)
                long long cp_last = wall_clock64();
#else
#define CP_MARK_NOWAIT(slot) do { } while (0)
#endif
                if (full_blocks) {
                    // Chunk operands in LDS, filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, lane l
                    // lands at base + 16 l bytes; no VGPR destination, no ds_write pass, half the vector-memory instructions
                    // of 8-byte loads), issued from INSIDE the hand-scheduled MFMA blocks of the chunk before (chunk_asm.h):
                    //   * X[j, k] (the other work-group's block, sc1): lanes 0..31 of a piece carry column 2 p, lanes 32..63
                    //     column 2 p + 1, two rows each -- the PAIR IMAGE  column c at (c >> 1) * 136 + (c & 1) * 64 doubles
                    //     (136 = 128 + 8: four columns apart = 16 (mod 32) 8-byte banks, as with S64_LS); wave w fills the
                    //     pairs w, w + 4, ..., alternately into sT0 / sT1;
                    //   * X[R, k] (own rows, plain): wave w's 16 rows x 8 columns per piece into ITS slice of sXI -- written
                    //     and read by that wave alone, so the fill of chunk k + 1 needs no barrier against the reads of k.
                    // A chunk: wait for its fills (vmcnt), own-row operand LDS -> registers (negated: T -= X B^T as 64
                    // MFMAs on -X), barrier (every wave's pieces of the B block are in; every wave is done with the other
                    // buffer), then the two MFMA halves carrying the fills of chunk k + 1.
                    const int pi = 4 * (i & 3) + (i >> 2);
                    const uint64_t st = (uint64_t)lda * 64u;                                 // 8 columns, bytes
                    const int wv = __builtin_amdgcn_readfirstlane(wave);
                    const uint32_t lbB0 = (uint32_t)(uintptr_t)sT0 + (uint32_t)wv * (CHUNK_PITCH * 8u);
                    const uint32_t lbB1 = (uint32_t)(uintptr_t)sT1 + (uint32_t)wv * (CHUNK_PITCH * 8u);
                    const uint32_t lbX = (uint32_t)(uintptr_t)sXI + (uint32_t)wv * (8u * CHUNK_PITCH * 8u);
                    const double* vaB = A + orow0 + 2 * (lane & 31) + (p.K0 + 2 * wave + (lane >> 5)) * lda;
                    const double* vaX = A + row0 + 16 * wave + 2 * (lane & 7) + (p.K0 + (lane >> 3)) * lda;
                    const double* myX = sXI + wave * (8 * CHUNK_PITCH) + (g >> 1) * CHUNK_PITCH + (4 * (g & 1)) * 16 + i;
                    chunk_dmaB(vaB, st, lbB0);
                    chunk_dmaX(vaX, st, lbX);
                    for (int k = 0; k < j; ++k) {
                        const bool more = (k + 1 < j);
                        CP_MARK_NOWAIT(4);
                        if (more && !wait_prog(o_prog, p.base + (unsigned long long)(k + 2), have, &s_seen, p.info, false, cb0 + Rr, cb0 + j)) return;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        CP_MARK_NOWAIT(0);
                        double xa[8], xb[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {              // column 16 I + 4 g + s: piece 2 I + (g >> 1), column 4 (g & 1) + s of it
                            xa[q] = -myX[(2 * (q >> 2)) * CHUNK_PITCH + (q & 3) * 16];
                            xb[q] = -myX[(4 + 2 * (q >> 2)) * CHUNK_PITCH + (q & 3) * 16];
                        }
                        __syncthreads();
                        CP_MARK_NOWAIT(1);
                        const uint32_t la = (uint32_t)(uintptr_t)((k & 1) ? sT1 : sT0) + (uint32_t)(((2 * g) * CHUNK_PITCH + pi) * 8);
                        if (more) {
                            vaB += (int64_t)PB * lda;
                            vaX += (int64_t)PB * lda;
                            chunk_half_dmaB(T, xa, la, vaB, st, (k & 1) ? lbB0 : lbB1);
                            CP_MARK_NOWAIT(2);
                            chunk_half_dmaX(T, xb, la + 16u * CHUNK_PITCH * 8u, vaX, st, lbX);
                            CP_MARK_NOWAIT(3);
                        } else {
                            chunk_half_plain(T, xa, la);
                            CP_MARK_NOWAIT(2);
                            chunk_half_plain(T, xb, la + 16u * CHUNK_PITCH * 8u);
                            CP_MARK_NOWAIT(3);
                        }
                    }
                } else {
                    double breg[16];
                    d4 XI[4];
                    load_block_regs(breg, A, lda, n, orow0, p.K0);
                    load_strip(XI, A, lda, n, row0, p.K0, PB);
                    for (int k = 0; k < j; ++k) {
                        double* sT = (k & 1) ? sT1 : sT0;
                        store_block_lds(breg, sT);
                        d4 XC[4];
#pragma unroll
                        for (int J = 0; J < 4; ++J) XC[J] = XI[J];
                        const bool more = (k + 1 < j);
                        if (more && !wait_prog(o_prog, p.base + (unsigned long long)(k + 2), have, &s_seen, p.info, false, cb0 + Rr, cb0 + j)) return;
                        __syncthreads();
                        if (more) {
                            const int64_t kc = p.K0 + (int64_t)(k + 1) * PB;
                            load_block_regs(breg, A, lda, n, orow0, kc);
                            load_strip(XI, A, lda, n, row0, kc, PB);
                        }
                        strip64_update(T, XC, sT);
                    }
                }
#ifdef PANEL_CHUNK_PROF
                if (p.trace && threadIdx.x == 0) {
#pragma unroll
                    for (int q_ = 0; q_ < 8; ++q_) p.trace[((int64_t)Rr * 40 + 16 + j) * 8 + q_] = cp[q_];
                }
#endif
            }
            // ---- X = T L_jj^{-T}
            PANEL_STAMP(1);
            if (Rr == j + 1 && jcols == PB && row0 + PB <= n) {
                // The pivot chain: this row block is the next diagonal owner.  L_jj arrives one 16-column block at
                // a time (potf2_64_lds raises qprog[j] behind block columns 0, 1, 2 and the row block's counter
                // behind the last; the inverse W_b of each 16 x 16 diagonal block comes with its column) and each
                // block is consumed while the owner still pivots the next one:
                //   X_b = T_b W_b^T,   T_J -= X_b L_Jb^T (J > b),   D -= X_b X_b^T
                // -- the MFMAs of solve64_lower_inv and strip64_update in the same order per accumulator.  The
                // loads of block b + 1 are issued as soon as its counter is up (one thread polls while the others
                // compute) and land during block b's diagonal update.  Behind the LAST block only its own stage is
                // left on the chain: load, 4 + 16 MFMAs, two barriers.
                double* sO = (sM == sT0) ? sT1 : sT0;     // X, transposed, for the diagonal update
                const unsigned long long* o_q = p.qprog + cb0 + j;
                const double* o_w = p.winv + (int64_t)(cb0 + j) * 1024;
                unsigned long long qhave = 0;
                if (!wait_prog(o_q, p.base + 1, qhave, &s_seen, p.info, true, cb0 + Rr, cb0 + j)) return;
                PANEL_STAMP(2);
                const int pi = 4 * (i & 3) + (i >> 2);
                const uint32_t svoff = (uint32_t)((16 * wave + i + (int64_t)(4 * g) * lda) * 8);
                double lv[4], wv;
                d4 Dd[4];
                strip_from_lds(Dd, sDD);
                load_lcol(lv, A, lda, orow0, col0, 0);
                wv = ldg_sc1(o_w, (uint32_t)(t * 8));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    stage_lcol(lv, sM, b);
                    sM[(16 * b + (t >> 4)) * S64_LS + 16 * b + (t & 15)] = wv;
                    __syncthreads();
                    double a[4];
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) a[s2] = sM[(16 * b + 4 * g + s2) * S64_LS + 16 * b + pi];
                    d4 Y = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], T[b][s2], Y, 0, 0, 0);
                    T[b] = Y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sO[(16 * b + 4 * g + r) * S64_LS + 16 * wave + i] = Y[r];
#pragma unroll
                    for (int J = b + 1; J < 4; ++J) {
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) a[s2] = -sM[(16 * b + 4 * g + s2) * S64_LS + 16 * J + pi];
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) T[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], T[b][s2], T[J], 0, 0, 0);
                    }
                    if (b < 3 && t == 64)
                        s_seen = (b < 2) ? poll_prog(o_q, p.base + (unsigned long long)(b + 2), p.info, cb0 + Rr, cb0 + j)
                                         : poll_prog(o_prog, p.base + (unsigned long long)(j + 1), p.info, cb0 + Rr, cb0 + j);
                    __syncthreads();
                    if (b < 3) {
                        if (s_seen == 0) return;          // the poll expired (uniform)
                        load_lcol(lv, A, lda, orow0, col0, b + 1);
                        wv = ldg_sc1(o_w + 256 * (b + 1), (uint32_t)(t * 8));
                        if (b == 2) PANEL_STAMP(4);
                    } else {
                        PANEL_STAMP(5);
                    }
#pragma unroll
                    for (int J = 0; J < 4; ++J) {
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) a[s2] = -sO[(16 * b + 4 * g + s2) * S64_LS + 16 * J + pi];
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) Dd[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], T[b][s2], Dd[J], 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg_off(A + row0 + (col0 + 16 * b + r) * lda, svoff, T[b][r]);
                }
                strip_to_lds(Dd, sDD);
                PANEL_STAMP(6);
                __syncthreads();                          // both buffers are free for potf2's scratch
                PANEL_STAMP(3);
                continue;
            }
            if (!wait_prog(o_prog, p.base + (unsigned long long)(j + 1), have, &s_seen, p.info, Rr == j + 1, cb0 + Rr, cb0 + j)) return;
            PANEL_STAMP(2);
            {
                double v[16];
                const int r = t & 63, cq = t >> 6;
                const int last = (int)(n - 1 - orow0);
                const uint32_t voff = (uint32_t)(((r <= last ? r : last) + (int64_t)cq * lda) * 8);
                if (jcols == PB) {                    // 16 plain loads in flight, masks applied afterwards
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = ldg_sc1(A + orow0 + (col0 + 4 * q) * lda, voff);
                } else {                              // ragged last block of the matrix
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int c = cq + 4 * q;
                        v[q] = ldg_sc1(A + orow0 + (r <= last ? r : last) + (col0 + (c < jcols ? c : 0)) * lda, 0u);
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = cq + 4 * q;
                    const bool ok = r >= c && r <= last && c < jcols;
                    v[q] = ok ? v[q] : ((r == c) ? 1.0 : 0.0);
                }
                // the diagonal 16 x 16 blocks arrive as their inverses, built once by L_jj's owner (potf2_64_lds)
                // instead of by every row block of the panel for itself
                const double* o_w = p.winv + (int64_t)(cb0 + j) * 1024;
                double wq[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) wq[b] = ldg_sc1(o_w + 256 * b, (uint32_t)(t * 8));
                __syncthreads();                  // every wave is done with sM (previous diagonal update)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = cq + 4 * q;
                    if ((r >> 4) != (c >> 4)) sM[c * S64_LS + r] = v[q];
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) sM[(16 * b + (t >> 4)) * S64_LS + 16 * b + (t & 15)] = wq[b];
            }
            __syncthreads();
            PANEL_STAMP(4);
            solve64_lower_inv(T, sM);
            PANEL_STAMP(5);
            store_strip(T, A, lda, n, row0, col0, jcols, false, is_diag);
            if (is_diag) {
                // D -= X X^T with X straight from the registers: the four strips meet in LDS
                __syncthreads();                      // all waves finished reading L_jj from sM
#pragma unroll
                for (int J = 0; J < 4; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sM[(16 * J + 4 * g + r) * S64_LS + 16 * wave + i] = T[J][r];
                __syncthreads();
                PANEL_STAMP(6);
                d4 Dd[4];
                strip_from_lds(Dd, sDD);
                strip64_update(Dd, T, sM);
                strip_to_lds(Dd, sDD);
                // X[R,j] is published only now: its global stores drained behind the MFMAs, so the release
                // fence is cheaper, and the barrier inside publish() also frees sM for its next use.  The
                // last one (j == Rr - 1) sits on the pivot chain: potf2_64_lds publishes it one barrier in.
                if (j + 1 < Rr) publish(my_prog, p.base + (unsigned long long)(j + 1));
                else __syncthreads();
            } else {
                __syncthreads();                      // the next step's first chunk may land in this buffer
            }
            PANEL_STAMP(3);
        }
        if (is_diag) {
            // ---- L_RR = chol(D): the accumulators go to LDS (identity-padded past a ragged end) and
            // potf2_64_lds factors them there, storing L to global on the way
            const int tstep = 39;
            PANEL_STAMP(0);
            double* sM = sDD;                         // D lives in LDS between the steps; potf2 factors it in place
            if (t < 64) sXT[t * POTF2_XS + 16] = 0.0;                       // potf2's hand-shake slots (potf2.h)
            __syncthreads();
            PANEL_STAMP(2);
            potf2_64_lds<S64_LS, true>(sM, sXT, A + row0 + row0 * lda, lda, dcols, (int)row0, p.info,
                                       Rr > 0 ? my_prog : nullptr, p.base + (unsigned long long)Rr,
                                       p.trace ? p.trace + ((int64_t)Rr * 40 + 30) * 8 : nullptr,
                                       p.qprog + cb0 + Rr, p.base, p.winv + (int64_t)(cb0 + Rr) * 1024, sT1);
            PANEL_STAMP(3);
            publish(my_prog, p.base + (unsigned long long)(Rr + 1));
            PANEL_STAMP(1);
        }
        __syncthreads();
    }
}


}  // namespace

size_t panel_ll_smem_bytes() { return (size_t)(4 * PB * S64_LS + PB * POTF2_XS) * sizeof(double); }

namespace {

int panel_workspaces(gpirt_handle_t h, hipStream_t stream, int64_t n)
{
    const int64_t need = (n + PB - 1) / PB + 1;
    if ((int64_t)h->prog_cap < need) {
        // (gpirt_create sizes the array for n <= 131008, so this is the rare path.)  Counters only ever grow
        // (epochs), so a fresh zeroed array is always consistent -- provided the fill is ORDERED before the launch:
        // it is issued on the launch stream and drained, never on the null stream (a hipStreamNonBlocking stream
        // does not wait for it).
        GP_HIP(hipStreamSynchronize(stream));
        if (h->side) GP_HIP(hipStreamSynchronize(h->side));
        if (h->stream != stream) GP_HIP(hipStreamSynchronize(h->stream));
        if (h->d_prog) GP_HIP(hipFree(h->d_prog));
        h->d_prog = nullptr;
        h->prog_cap = 0;
        GP_HIP(hipMalloc(&h->d_prog, 2 * (size_t)need * sizeof(unsigned long long)));      // prog | qprog
        GP_HIP(hipMemsetAsync(h->d_prog, 0, 2 * (size_t)need * sizeof(unsigned long long), stream));
        GP_HIP(hipStreamSynchronize(stream));
        h->prog_cap = (size_t)need;
    }
    if (h->winv_blocks < need) {            // W_b: one 4 x 256 slot per 64-column block of the matrix
        GP_HIP(hipStreamSynchronize(stream));
        if (h->side) GP_HIP(hipStreamSynchronize(h->side));
        if (h->d_winv) GP_HIP(hipFree(h->d_winv));
        h->d_winv = nullptr; h->winv_blocks = 0;
        const int64_t blocks = need < 160 ? 160 : need;
        GP_HIP(hipMalloc(&h->d_winv, (size_t)blocks * 1024 * sizeof(double)));
        h->winv_blocks = blocks;
    }
    if (!h->panel_attr_set) {
        GP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_ll_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_ll_smem_bytes()));
        h->panel_attr_set = true;
    }
    return 0;
}

}  // namespace

// Factor panel columns [K0, c1) with the persistent kernel, sweeping the rows [K0, row_end) (row_end <= 0: all n rows).
// Opens a new epoch of the progress counters.
int launch_panel_ll(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t K0, int64_t c1,
                    int64_t row_end, unsigned long long* epoch_out)
{
    if (K0 >= c1) return 0;
    GP_TRY(panel_workspaces(h, stream, n));
    const int n_cu = h->n_cu;
    PanelArgs p;
    p.A = A; p.lda = lda; p.n = n; p.K0 = K0; p.c1 = c1;
    p.prog = h->d_prog;
    p.qprog = h->d_prog + h->prog_cap;
    p.winv = h->d_winv;
    h->prog_seq += 1;
    p.base = h->prog_seq * 64ull;                    // a panel publishes at most ncb + 1 <= 17 steps
    if (epoch_out) *epoch_out = p.base;
    p.info = h->d_info;
    p.trace = (h->panel_trace && (h->panel_trace_k0 < 0 || h->panel_trace_k0 == K0)) ? h->panel_trace : nullptr;
    p.nrb = (int)((n - K0 + PB - 1) / PB);
    p.ncb = (int)((c1 - K0 + PB - 1) / PB);
    if (p.ncb > 32) { set_error("panel wider than 2048 columns"); return GPIRT_E_ARG; }
    p.rb_begin = 0;
    p.rb_end = p.nrb;
    if (row_end > 0 && row_end < n) {
        if (row_end < c1 || (row_end - K0) % PB != 0) { set_error("restricted panel launch: row_end must cover the diagonal blocks in whole 64-row blocks"); return GPIRT_E_ARG; }
        p.rb_end = (int)((row_end - K0) / PB);
    }
    const int nb = p.rb_end - p.rb_begin;
    const int grid = nb < n_cu ? nb : n_cu;
    hipLaunchKernelGGL(panel_ll_kernel, dim3((unsigned)grid), dim3(256), panel_ll_smem_bytes(), stream, p);
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
