// trsm.hip -- triangular solves with many right-hand sides against the shared Cholesky factor:
//   trans = 0 :  L   X = B   solve(trimatl(L), .)      src/draw-fstar.cpp:19 and :7 (inner)
//   trans = 1 :  L^T X = B   solve(trimatu(L.t()), .)  src/draw-fstar.cpp:7 (outer)
// The reference runs these as LAPACK dtrtrs one vector at a time (2 m solves per iteration, L
// re-read 2 m times); here all item columns go through one recursive blocked solve:
//   split the rows in two, solve the first half, fold it into the second half with ONE fp64-MFMA
//   gemm (K = half the rows, so the deep products carry the flops), recurse.  The leaves run the
//   MFMA-layout substitution of solve64.h: off-diagonal 16 x 16 coupling by MFMA, the 16 x 16 diagonal
//   blocks through their inverses (formed on the fly in LDS).  The 512-row leaves of the recursion apply explicit
//   inverses of L's 512 x 512 diagonal blocks (1024 x 1024 for thin solves), built once per factor: only DIAGONAL
//   BLOCKS OF L are ever inverted -- their condition is bounded by cond(L) = sqrt(cond(S)) ~ 1e3.
#include "common.h"
#include "kernels.h"
#include "solve64.h"

#include <stdlib.h>

namespace gpirt {

namespace {

constexpr int NL = 64;          // leaf rows
constexpr int CB = 64;          // right-hand-side columns per work-group

// A 64 x 64 coefficient block goes global -> registers -> LDS:
//   s[c * S64_LS + r] = M[64 rb + r][64 cb + c];  M = L (forward) or the flipped transpose (backward).
template <bool BACK>
__device__ __forceinline__ void stage_block(const double* __restrict__ L, int64_t ldl, int nb, int rb, int cb,
                                            bool diag, double* __restrict__ s)
{
    // All 16 loads of a thread are issued before the first LDS store (one memory round trip, not 16).
    // (Fetching the next block into registers while this one computes was measured twice -- one and two
    // blocks ahead -- and made every stage slower, not faster; the blocks come from L2 anyway.)
    const int t = threadIdx.x;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = t + 256 * k;
        int r, c;
        if (!BACK) { r = idx & (NL - 1); c = idx >> 6; } else { c = idx & (NL - 1); r = idx >> 6; }
        const int a = 64 * rb + r, b = 64 * cb + c;            // M[a][b]
        const bool in = a < nb && b < nb;
        const int ac = in ? a : 0, bc = in ? b : 0;            // clamped: the load itself is unconditional
        v[k] = BACK ? L[(int64_t)(nb - 1 - bc) + (int64_t)(nb - 1 - ac) * ldl] : L[(int64_t)ac + (int64_t)bc * ldl];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = t + 256 * k;
        int r, c;
        if (!BACK) { r = idx & (NL - 1); c = idx >> 6; } else { c = idx & (NL - 1); r = idx >> 6; }
        const int a = 64 * rb + r, b = 64 * cb + c;
        const bool in = a < nb && b < nb && (!diag || a >= b);
        s[c * S64_LS + r] = in ? v[k] : ((diag && a == b) ? 1.0 : 0.0);   // identity padding of a ragged last block
    }
}

// Solves the nb x nb system for 64 right-hand sides per work-group (16 per wavefront) with the
// MFMA-layout substitution core.  BACK = false: L x = b.  BACK = true: L^T x = b, run as the same
// forward substitution on the flipped transpose M'[a][b] = L[nb-1-b][nb-1-a] with the vector
// entries visited in reverse order.
template <bool BACK>
__global__ __launch_bounds__(256) void trsm_leaf_kernel(const double* __restrict__ L, int64_t ldl,
                                                        int nb, double* __restrict__ B, int64_t ldb,
                                                        int64_t nrhs)
{
    __shared__ __attribute__((aligned(16))) double sM[NL * S64_LS];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int64_t col = (int64_t)blockIdx.x * CB + wave * 16 + i;
    const bool live = col < nrhs;
    double* b = B + col * ldb;
    d4 X[4];
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * J + 4 * g + r;
            const int p = BACK ? (nb - 1 - c) : c;
            X[J][r] = (live && c < nb) ? b[p] : 0.0;
        }
    stage_block<BACK>(L, ldl, nb, 0, 0, true, sM);
    __syncthreads();
    invert_diag16(sM);
    solve64_lower_inv(X, sM);
    if (live) {
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * J + 4 * g + r;
                const int p = BACK ? (nb - 1 - c) : c;
                if (c < nb) b[p] = X[J][r];
            }
    }
}

// Fused 256-row leaf: one work-group carries 64 right-hand sides (16 per wavefront) through up to
// four 64-row blocks without leaving the chip: the vectors stay in registers (16 x d4 per lane),
// every 64 x 64 coefficient block of L is staged once through LDS (double buffered), off-diagonal
// blocks are applied with MFMA (strip64_update), diagonal blocks with solve64_lower.  Replaces
// 4 leaf launches + 3 small gemm launches (and their 6 kernel boundaries) of the recursion.
constexpr int NL4 = 256;

// IDENT: the right-hand side is the identity (nothing is read from B) and blockIdx.y walks a batch of
// diagonal blocks -- L advances by lstride per entry, B by bstride per PAIR of entries plus bhalf for the odd
// one (the two diagonal quarters of a 512 x 512 inverse): the block inverses of trsm_rec.
// OCC = 2: the same code held to 256 registers (269 of them spilled to scratch): it then fits BESIDE a work-group of the
// 128-tile product (224 registers) on a CU -- the block inverses of the draw_fstar side chain, which start beside
// nu = L z (sampler.hip); at 392 registers the kernel waited for that product's work-groups to leave (1.09 ms).
template <bool BACK, bool IDENT = false, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void trsm_leaf256_kernel(const double* __restrict__ L, int64_t ldl, int nb,
                                                           double* __restrict__ B, int64_t ldb, int64_t nrhs,
                                                           long long* trace, int64_t lstride = 0, int64_t bstride = 0,
                                                           int64_t bhalf = 0)
{
    L += (int64_t)blockIdx.y * lstride;
    B += (int64_t)(blockIdx.y >> 1) * bstride + (int64_t)(blockIdx.y & 1) * bhalf;
    // trace: optional 100 MHz stamps of work-group 0 (tools/micro/leaf_bench.hip), nullptr in the library
#define LEAF_STAMP(slot) do { if (trace && blockIdx.x == 0 && threadIdx.x == 0) trace[slot] = wall_clock64(); } while (0)
    LEAF_STAMP(0);
    __shared__ __attribute__((aligned(16))) double sbuf[2][NL * S64_LS];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nblk = (nb + NL - 1) / NL;
    const int64_t col0 = (int64_t)blockIdx.x * CB;
    // The right-hand sides move between memory and the MFMA layout (lane = column, registers = rows) through
    // LDS, 64 x 64 tiles at a time, so every global access runs along a column of B: 512 contiguous bytes per
    // wavefront instead of 64 lanes x 8 bytes scattered over 64 cache lines.
    //   tile element (e, cc): solve-order entry 64 JB + e of right-hand side col0 + cc, kept at s[cc * LS + e]
    const int te = t & 63, tcq = t >> 6;               // this thread moves entries te of columns tcq + 4 k
    d4 X[16];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        double xv[2][16];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int JB = 2 * half + h2;
            const int c = 64 * JB + te;                // solve-order entry
            const int prow = BACK ? (nb - 1 - c) : c;  // its row in memory
            if (IDENT) {
#pragma unroll
                for (int k = 0; k < 16; ++k) xv[h2][k] = ((int64_t)c == col0 + tcq + 4 * k) ? 1.0 : 0.0;
            } else if (64 * JB + 64 <= nb && col0 + CB <= nrhs) {   // interior tile (uniform): plain loads, all in flight
                const double* src = B + prow + (col0 + tcq) * ldb;
#pragma unroll
                for (int k = 0; k < 16; ++k) xv[h2][k] = src[(int64_t)(4 * k) * ldb];
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int64_t cc = col0 + tcq + 4 * k;
                    const bool ok = (c < nb) && (cc < nrhs);
                    const double x = B[(int64_t)(ok ? prow : 0) + (ok ? cc : col0) * ldb];
                    xv[h2][k] = ok ? x : 0.0;
                }
            }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int k = 0; k < 16; ++k) sbuf[h2][(tcq + 4 * k) * S64_LS + te] = xv[h2][k];
        __syncthreads();
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 lo = *reinterpret_cast<const double2*>(&sbuf[h2][(16 * wave + i) * S64_LS + 16 * q + 4 * g]);
                const double2 hi = *reinterpret_cast<const double2*>(&sbuf[h2][(16 * wave + i) * S64_LS + 16 * q + 4 * g + 2]);
                X[8 * half + 4 * h2 + q] = d4{ lo.x, lo.y, hi.x, hi.y };
            }
        __syncthreads();
    }
    // the (JB, IB) stages in execution order: row block JB takes its off-diagonal blocks IB < JB, then its
    // diagonal block
    int buf = 0;
    int stamp = 1;
#pragma unroll
    for (int JB = 0; JB < 4; ++JB) {
        if (JB < nblk) {
#pragma unroll
            for (int IB = 0; IB <= JB; ++IB) {
                const bool diag = (IB == JB);
                stage_block<BACK>(L, ldl, nb, JB, IB, diag, sbuf[buf]);
                __syncthreads();
                LEAF_STAMP(stamp); ++stamp;
                if (!diag) {
                    strip64_update(reinterpret_cast<d4(&)[4]>(X[4 * JB]), reinterpret_cast<const d4(&)[4]>(X[4 * IB]),
                                   sbuf[buf]);
                } else {
                    invert_diag16(sbuf[buf]);
                    solve64_lower_inv(reinterpret_cast<d4(&)[4]>(X[4 * JB]), sbuf[buf]);
                }
                buf ^= 1;
            }
        }
    }
    LEAF_STAMP(stamp); ++stamp;
    __syncthreads();                                   // the last stage has finished with both buffers
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const d4 x = X[8 * half + 4 * h2 + q];
                *reinterpret_cast<double2*>(&sbuf[h2][(16 * wave + i) * S64_LS + 16 * q + 4 * g]) = double2{ x[0], x[1] };
                *reinterpret_cast<double2*>(&sbuf[h2][(16 * wave + i) * S64_LS + 16 * q + 4 * g + 2]) = double2{ x[2], x[3] };
            }
        __syncthreads();
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int JB = 2 * half + h2;
            const int c = 64 * JB + te;
            const int prow = BACK ? (nb - 1 - c) : c;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t cc = col0 + tcq + 4 * k;
                if (c < nb && cc < nrhs) B[(int64_t)prow + cc * ldb] = sbuf[h2][(tcq + 4 * k) * S64_LS + te];
            }
        }
        __syncthreads();
    }
    LEAF_STAMP(stamp);
#undef LEAF_STAMP
}

// rows of B <- the rows x nrhs product parked in the workspace (ld = rows; rows = 256 or 512)
__global__ __launch_bounds__(256) void copy_back_kernel(const double* __restrict__ T, int rows, double* __restrict__ B,
                                                        int64_t ldb, int64_t nrhs)
{
    const int64_t c = blockIdx.x;
    if (c >= nrhs) return;
    for (int r = threadIdx.x; r < rows; r += 256) B[r + c * ldb] = T[r + c * (int64_t)rows];
}

constexpr int NI = 512;         // edge of the explicitly inverted diagonal blocks
constexpr int NQ = 1024;        // ... and of the larger ones built from pairs of them for thin solves

// W1024 slot q (ld 1024): the two 512 x 512 inverses on the diagonal, zeros above; the lower-left quarter is
// written afterwards by the batched products  -W_b (L_ba W_a)
__global__ __launch_bounds__(256) void assemble_quad_kernel(const double* __restrict__ W512, double* __restrict__ W1024)
{
    const int64_t q = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over 1024 x 512: column c, rows of one half
    const int c = (int)(idx / NQ), r = (int)(idx % NQ);
    if (c >= NQ) return;
    double* dst = W1024 + q * (int64_t)NQ * NQ + r + (int64_t)c * NQ;
    if (c < NI) {
        if (r < NI) *dst = W512[(2 * q) * (int64_t)NI * NI + r + (int64_t)c * NI];
    } else {
        *dst = (r < NI) ? 0.0 : W512[(2 * q + 1) * (int64_t)NI * NI + (r - NI) + (int64_t)(c - NI) * NI];
    }
}

// Forward solves with winv != nullptr: the inverses of L's diagonal blocks, 512 x 512 block b at
// winv + b * 512 * 512 (ld 512, lower triangle; npair of them), plus -- when an odd full 256-row block is left
// over -- its 256 x 256 inverse in slot npair.  A 512-aligned leaf is then the product W_b * B_b on all CUs
// (lower-triangular GEMM into the workspace + copy back, ~44 us) instead of two 48 us substitutions on
// nrhs / 64 of them with a GEMM in between; 256-row leaves use the diagonal quarters the same way.
int trsm_rec(gpirt_handle_t h, hipStream_t stream, const double* L, int64_t ldl, double* B,
             int64_t nrhs, int64_t ldb, bool trans, int64_t r0, int64_t r1, const double* winv = nullptr,
             int64_t npair = 0, bool odd = false, const double* wquad = nullptr, int64_t nquad = 0)
{
    const int64_t len = r1 - r0;
    if (wquad && len == NQ && (r0 % NQ) == 0 && r0 / NQ < nquad &&
        gemm_split_count(h, stream, TRI_NONE, len, nrhs, len) >= 2) {
        // thin solve: one product with the 1024 x 1024 inverse (split over K, so it may land in B itself)
        return launch_gemm(h, stream, trans, false, TRI_NONE, len, nrhs, len, 1.0, wquad + (r0 / NQ) * (int64_t)NQ * NQ, NQ,
                           B + r0, ldb, 0.0, B + r0, ldb);
    }
    if (winv) {
        const double* W = nullptr;
        if (len == NI && (r0 % NI) == 0 && r0 / NI < npair) {
            W = winv + (r0 / NI) * (int64_t)(NI * NI);
        } else if (len == NL4 && (r0 % NL4) == 0) {
            if (r0 / NI < npair) W = winv + (r0 / NI) * (int64_t)(NI * NI) + ((r0 % NI) ? (int64_t)NL4 * (NI + 1) : 0);
            else if (odd && r0 == npair * NI) W = winv + npair * (int64_t)(NI * NI);
        }
        if (W) {
            // few right-hand sides: W_b as a dense operand (its upper-right quarter is zero), so the product is
            // eligible for the split-K path of launch_gemm; many: skip the structural zeros instead
            // (the transposed solve multiplies by W_b^T: the same stored block, read transposed)
            const int tri = (trans || nrhs <= 1280) ? TRI_NONE : TRI_A_LOWER;
            if (gemm_split_count(h, stream, tri, len, nrhs, len) >= 2) {
                // split over K: the parts are complete before the sum is written, so it can land in B itself
                return launch_gemm(h, stream, trans, false, tri, len, nrhs, len, 1.0, W, NI, B + r0, ldb, 0.0, B + r0, ldb);
            }
            GP_TRY(launch_gemm(h, stream, trans, false, tri, len, nrhs, len, 1.0, W, NI, B + r0, ldb, 0.0,
                               h->d_trsm_tmp, len));
            hipLaunchKernelGGL(copy_back_kernel, dim3((unsigned)nrhs), dim3(256), 0, stream, h->d_trsm_tmp, (int)len,
                               B + r0, ldb, nrhs);
            return 0;
        }
    }
    if (len <= NL4 && len > NL) {
        const unsigned grid = (unsigned)((nrhs + CB - 1) / CB);
        if (!trans)
            hipLaunchKernelGGL(trsm_leaf256_kernel<false>, dim3(grid), dim3(256), 0, stream,
                               L + r0 + r0 * ldl, ldl, (int)len, B + r0, ldb, nrhs, (long long*)nullptr);
        else
            hipLaunchKernelGGL(trsm_leaf256_kernel<true>, dim3(grid), dim3(256), 0, stream,
                               L + r0 + r0 * ldl, ldl, (int)len, B + r0, ldb, nrhs, (long long*)nullptr);
        return 0;
    }
    if (len <= NL) {
        const unsigned grid = (unsigned)((nrhs + CB - 1) / CB);
        if (!trans)
            hipLaunchKernelGGL(trsm_leaf_kernel<false>, dim3(grid), dim3(256), 0, stream,
                               L + r0 + r0 * ldl, ldl, (int)len, B + r0, ldb, nrhs);
        else
            hipLaunchKernelGGL(trsm_leaf_kernel<true>, dim3(grid), dim3(256), 0, stream,
                               L + r0 + r0 * ldl, ldl, (int)len, B + r0, ldb, nrhs);
        return 0;
    }
    // split at a multiple of the fused-leaf height nearest the middle
    int64_t half = ((len / 2 + NL4 - 1) / NL4) * NL4;
    if (half >= len) half = ((len / 2 + NL - 1) / NL) * NL;
    const int64_t mid = r0 + half;
    if (!trans) {
        GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, r0, mid, winv, npair, odd, wquad, nquad));
        // B[mid:r1, :] -= L[mid:r1, r0:mid] * B[r0:mid, :]
        GP_TRY(launch_gemm(h, stream, false, false, TRI_NONE, r1 - mid, nrhs, mid - r0, -1.0,
                           L + mid + r0 * ldl, ldl, B + r0, ldb, 1.0, B + mid, ldb));
        GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, mid, r1, winv, npair, odd, wquad, nquad));
    } else {
        GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, mid, r1, winv, npair, odd, wquad, nquad));
        // B[r0:mid, :] -= L[mid:r1, r0:mid]^T * B[mid:r1, :]
        GP_TRY(launch_gemm(h, stream, true, false, TRI_NONE, mid - r0, nrhs, r1 - mid, -1.0,
                           L + mid + r0 * ldl, ldl, B + mid, ldb, 1.0, B + r0, ldb));
        GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, r0, mid, winv, npair, odd, wquad, nquad));
    }
    return 0;
}

}  // namespace

// Workspaces for the block inverses of an n x n factor (512-blocks; 1024-blocks too when `thin`) and the product
// buffer; grown on demand (synchronises `stream` when it reallocates).
int trsm_inverses_reserve(gpirt_handle_t h, hipStream_t stream, int64_t n, int64_t nrhs, bool thin)
{
    const int64_t nfull = n / NL4, npair = nfull / 2;
    const size_t wbytes = (size_t)(npair + 1) * NI * NI * sizeof(double);
    size_t tbytes = (size_t)NI * (size_t)nrhs * sizeof(double);
    if (tbytes < (size_t)npair * NL4 * NL4 * sizeof(double)) tbytes = (size_t)npair * NL4 * NL4 * sizeof(double);
    if (thin && tbytes < (size_t)(npair / 2) * NI * NI * sizeof(double)) tbytes = (size_t)(npair / 2) * NI * NI * sizeof(double);
    if (h->trsm_winv_bytes < wbytes || h->trsm_tmp_bytes < tbytes) {
        GP_HIP(hipStreamSynchronize(stream));
        if (h->trsm_winv_bytes < wbytes) {
            if (h->d_trsm_winv) GP_HIP(hipFree(h->d_trsm_winv));
            h->d_trsm_winv = nullptr; h->trsm_winv_bytes = 0;
            GP_HIP(hipMalloc(&h->d_trsm_winv, wbytes));
            GP_HIP(hipMemsetAsync(h->d_trsm_winv, 0, wbytes, stream));   // upper-right quarters stay zero for good
            GP_HIP(hipStreamSynchronize(stream));
            h->trsm_winv_bytes = wbytes;
        }
        if (h->trsm_tmp_bytes < tbytes) {
            if (h->d_trsm_tmp) GP_HIP(hipFree(h->d_trsm_tmp));
            h->d_trsm_tmp = nullptr; h->trsm_tmp_bytes = 0;
            GP_HIP(hipMalloc(&h->d_trsm_tmp, tbytes));
            h->trsm_tmp_bytes = tbytes;
        }
    }
    if (thin && npair >= 2) {
        const size_t qbytes = (size_t)(npair / 2) * NQ * NQ * sizeof(double);
        if (h->trsm_wquad_bytes < qbytes) {
            GP_HIP(hipStreamSynchronize(stream));
            if (h->d_trsm_wquad) GP_HIP(hipFree(h->d_trsm_wquad));
            h->d_trsm_wquad = nullptr; h->trsm_wquad_bytes = 0;
            GP_HIP(hipMalloc(&h->d_trsm_wquad, qbytes));
            h->trsm_wquad_bytes = qbytes;
        }
    }
    return 0;
}

// Inverses of L's diagonal blocks for the 512-block pairs [p0, p1) (p = index of a 512 x 512 diagonal block; the odd
// 256-row block behind the last pair rides with the last range), into h->d_trsm_winv / d_trsm_wquad:
//  1. the full 256 x 256 diagonal blocks: ONE batched launch of the fused leaf on identity right-hand sides
//     (4 work-groups per block), written straight into the diagonal quarters of the 512 x 512 slots;
//  2. the lower-left quarter of each slot,  -W2 (L21 W1),  as two batched 256^3 MFMA products;
//  3. thin solves (few right-hand sides are launch-bound, not flop-bound): pairs of 512-blocks merged into 1024 x 1024
//     inverses (two more batched products), which halves the leaves and drops a recursion level.
// A range only reads the diagonal blocks it inverts, so the blocks of every outer panel but the last are built while the
// LAST outer panel is still being factored -- a phase that leaves most of the chip idle -- on the sampler's own stream
// (sampler.hip, do_factor; round 3), and only the last range + the solve follow the factorisation.  p0 must be even when
// `thin`.  tests/test_gpu_ops.py::test_trsm_inverses_piecewise compares a piecewise build with the one-range build.
int trsm_inverses_build(gpirt_handle_t h, hipStream_t stream, const double* L, int64_t n, int64_t ldl, bool thin,
                        int64_t p0, int64_t p1)
{
    const int64_t nfull = n / NL4, npair = nfull / 2;
    if (p1 > npair) p1 = npair;
    if (p0 >= p1 && !(p1 == npair && (nfull & 1))) return 0;
    double* W = h->d_trsm_winv;
    // 256-blocks 2 p0 .. 2 p1 - 1 (+ the odd one when this range closes the matrix)
    const int64_t b0 = 2 * p0, b1 = (p1 == npair) ? nfull : 2 * p1;
    if (b1 > b0) {
        if (h->slim_leaf)
            hipLaunchKernelGGL((trsm_leaf256_kernel<false, true, 2>), dim3(NL4 / CB, (unsigned)(b1 - b0)), dim3(256), 0, stream,
                               L + b0 * (int64_t)NL4 * (ldl + 1), ldl, NL4, W + p0 * (int64_t)NI * NI, (int64_t)NI, (int64_t)NL4,
                               (long long*)nullptr, (int64_t)NL4 * (ldl + 1), (int64_t)NI * NI, (int64_t)NL4 * (NI + 1));
        else
            hipLaunchKernelGGL((trsm_leaf256_kernel<false, true>), dim3(NL4 / CB, (unsigned)(b1 - b0)), dim3(256), 0, stream,
                               L + b0 * (int64_t)NL4 * (ldl + 1), ldl, NL4, W + p0 * (int64_t)NI * NI, (int64_t)NI, (int64_t)NL4,
                               (long long*)nullptr, (int64_t)NL4 * (ldl + 1), (int64_t)NI * NI, (int64_t)NL4 * (NI + 1));
    }
    const int np = (int)(p1 - p0);
    if (np > 0) {
        const double* Lp = L + p0 * (int64_t)NI * (ldl + 1);
        double* Wp = W + p0 * (int64_t)NI * NI;
        double* Tp = h->d_trsm_tmp + p0 * (int64_t)NL4 * NL4;
        // T_b = L21 W1
        GP_TRY(launch_gemm_batched(stream, false, false, TRI_NONE, NL4, NL4, NL4, 1.0,
                                   Lp + NL4, ldl, (int64_t)NI * (ldl + 1), Wp, NI, (int64_t)NI * NI,
                                   0.0, Tp, NL4, (int64_t)NL4 * NL4, np));
        // W21 = -W2 T_b
        GP_TRY(launch_gemm_batched(stream, false, false, TRI_A_LOWER, NL4, NL4, NL4, -1.0,
                                   Wp + (int64_t)NL4 * (NI + 1), NI, (int64_t)NI * NI, Tp, NL4,
                                   (int64_t)NL4 * NL4, 0.0, Wp + NL4, NI, (int64_t)NI * NI, np));
    }
    if (thin && npair >= 2) {
        const int64_t q0 = p0 / 2, q1 = (p1 / 2 < npair / 2) ? p1 / 2 : npair / 2;
        const int nq = (int)(q1 - q0);
        if (nq > 0) {
            double* Wq = h->d_trsm_wquad + q0 * (int64_t)NQ * NQ;
            const double* Lq = L + q0 * (int64_t)NQ * (ldl + 1);
            double* Tq = h->d_trsm_tmp + q0 * (int64_t)NI * NI;
            hipLaunchKernelGGL(assemble_quad_kernel, dim3((unsigned)((int64_t)NQ * NQ / 256), (unsigned)nq), dim3(256), 0, stream,
                               W + 2 * q0 * (int64_t)NI * NI, Wq);
            // T = L_ba W_a (512^3, W_a lower triangular), then the lower-left quarter  -W_b T
            GP_TRY(launch_gemm_batched(stream, false, false, TRI_NONE, NI, NI, NI, 1.0,
                                       Lq + NI, ldl, (int64_t)NQ * (ldl + 1), Wq, NQ, (int64_t)NQ * NQ,
                                       0.0, Tq, NI, (int64_t)NI * NI, nq));
            GP_TRY(launch_gemm_batched(stream, false, false, TRI_A_LOWER, NI, NI, NI, -1.0,
                                       Wq + (int64_t)NI * (NQ + 1), NQ, (int64_t)NQ * NQ, Tq, NI,
                                       (int64_t)NI * NI, 0.0, Wq + NI, NQ, (int64_t)NQ * NQ, nq));
        }
    }
    GP_HIP(hipGetLastError());
    return 0;
}

void trsm_inverses_mark(gpirt_handle_t h, const double* L, int64_t n, int64_t ldl, bool thin)
{
    const int64_t npair = (n / NL4) / 2;
    h->trsm_winv_L = L; h->trsm_winv_n = n; h->trsm_winv_ld = ldl;
    h->trsm_quads = (thin && npair >= 2) ? npair / 2 : 0;
}

// reuse_inverses: the block inverses built by the previous call on this handle are still those of L (same L,
// n and ldl, not modified since) -- the sampler's second solve of a draw_fstar skips rebuilding them.
int launch_trsm_lower(gpirt_handle_t h, hipStream_t stream, const double* L, int64_t n, int64_t ldl,
                      double* B, int64_t nrhs, int64_t ldb, bool trans, bool reuse_inverses)
{
    if (n <= 0 || nrhs <= 0) return 0;
    // GPIRT_TRSM_INV=2 keeps every leaf a substitution (tests/test_gpu_configs.py sets it through gpirt_config_set to
    // price the block inverses against LAPACK)
    const bool use_inv = h->cfg.trsm_inv != 2;
    const double* winv = nullptr;
    const int64_t nfull = n / NL4, npair = nfull / 2;
    const bool odd = (nfull & 1) != 0;
    // "thin": the 1024 x 1024 inverses are built too and a 1024-aligned leaf is ONE product with them where launch_gemm would
    // split that product over K (trsm_rec: up to 1280 right-hand sides) -- half the leaves and one recursion level less.
    // Round 5: not only for the rank-r draw_fstar's 64 columns but for the m item columns of the other forms too
    // (8192 x 1024: fused 9.30 -> 9.16 ms per iteration, as written 10.96 -> 10.66)
    const bool thin = nrhs <= 1280;
    const bool have = reuse_inverses && h->trsm_winv_L == L && h->trsm_winv_n == n && h->trsm_winv_ld == ldl;
    if (use_inv && nfull >= 2 && nrhs >= 64 && have) {
        winv = h->d_trsm_winv;
    } else if (use_inv && nfull >= 2 && nrhs >= 64) {
        GP_TRY(trsm_inverses_reserve(h, stream, n, nrhs, thin));
        GP_TRY(trsm_inverses_build(h, stream, L, n, ldl, thin, 0, npair));
        winv = h->d_trsm_winv;
        h->trsm_winv_L = L; h->trsm_winv_n = n; h->trsm_winv_ld = ldl;
        h->trsm_quads = (thin && npair >= 2) ? npair / 2 : 0;
    }
    if (winv) {
        // the leaf products park a 512 x nrhs block in the workspace
        const size_t tb = (size_t)NI * (size_t)nrhs * sizeof(double);
        if (h->trsm_tmp_bytes < tb) {
            GP_HIP(hipStreamSynchronize(stream));
            if (h->d_trsm_tmp) GP_HIP(hipFree(h->d_trsm_tmp));
            h->d_trsm_tmp = nullptr; h->trsm_tmp_bytes = 0;
            GP_HIP(hipMalloc(&h->d_trsm_tmp, tb));
            h->trsm_tmp_bytes = tb;
        }
    }
    const bool use_quads = winv && thin && h->trsm_quads > 0;
    GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, 0, n, winv, npair, odd, use_quads ? h->d_trsm_wquad : nullptr,
                    use_quads ? h->trsm_quads : 0));
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
