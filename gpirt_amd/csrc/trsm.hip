// trsm.hip -- triangular solves with many right-hand sides against the shared Cholesky factor:
//   trans = 0 :  L   X = B   solve(trimatl(L), .)      src/draw-fstar.cpp:19 and :7 (inner)
//   trans = 1 :  L^T X = B   solve(trimatu(L.t()), .)  src/draw-fstar.cpp:7 (outer)
// The reference runs these as LAPACK dtrtrs one vector at a time (2 m solves per iteration, L
// re-read 2 m times); here all item columns go through one recursive blocked solve:
//   split the rows in two, solve the first half, fold it into the second half with ONE fp64-MFMA
//   gemm (K = half the rows, so the deep products carry the flops), recurse.  The 64-row leaves
//   are true substitutions (no inverted blocks: S has condition ~1e6 by construction):
//   one lane owns one right-hand-side column, its 64 entries live in registers, the 64 x 64
//   diagonal block of L is broadcast from LDS, and the tile of B is transposed through LDS so
//   global traffic stays coalesced.
#include "common.h"
#include "kernels.h"

namespace gpirt {

namespace {

constexpr int NL = 64;          // leaf rows
constexpr int CB = 64;          // right-hand-side columns per work-group
constexpr int LDT = NL + 1;

// Solves the nb x nb system for 64 columns.  BACK = false: L x = b ; BACK = true: L^T x = b.
template <bool BACK>
__global__ __launch_bounds__(256) void trsm_leaf_kernel(const double* __restrict__ L, int64_t ldl,
                                                        int nb, double* __restrict__ B, int64_t ldb,
                                                        int64_t nrhs)
{
    // sL[c][c2]: for !BACK element L[c2][c] (column c of L, rows c2 >= c);
    //            for  BACK element L[c][c2] (row c of L, columns c2 <= c).  Identity padded.
    __shared__ __attribute__((aligned(16))) double sL[NL * NL];
    __shared__ double sT[CB * LDT];     // sT[col][row]
    const int t = threadIdx.x;
    for (int idx = t; idx < NL * NL; idx += 256) {
        const int r = idx & (NL - 1), c = idx >> 6;      // L[r][c] with r >= c is stored data
        double v = 0.0;
        if (r < nb && c < nb && r >= c) v = L[(int64_t)r + (int64_t)c * ldl];
        else if (r == c) v = 1.0;
        if (!BACK) sL[c * NL + r] = v; else sL[r * NL + c] = v;
    }
    const int64_t col0 = (int64_t)blockIdx.x * CB;
    // coalesced tile load: 64 rows x 64 columns
    for (int idx = t; idx < NL * CB; idx += 256) {
        const int r = idx & (NL - 1), c = idx >> 6;
        double v = 0.0;
        if (r < nb && col0 + c < nrhs) v = B[(int64_t)r + (col0 + c) * ldb];
        sT[c * LDT + r] = v;
    }
    __syncthreads();
    if (t < CB) {
        double x[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) x[i] = sT[t * LDT + i];
        if (!BACK) {
#pragma unroll
            for (int c = 0; c < NL; ++c) {
                const double xc = x[c] / sL[c * NL + c];
                x[c] = xc;
#pragma unroll
                for (int c2 = c + 1; c2 < NL; ++c2) x[c2] -= xc * sL[c * NL + c2];
            }
        } else {
#pragma unroll
            for (int c = NL - 1; c >= 0; --c) {
                const double xc = x[c] / sL[c * NL + c];
                x[c] = xc;
#pragma unroll
                for (int c2 = 0; c2 < c; ++c2) x[c2] -= xc * sL[c * NL + c2];
            }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) sT[t * LDT + i] = x[i];
    }
    __syncthreads();
    for (int idx = t; idx < NL * CB; idx += 256) {
        const int r = idx & (NL - 1), c = idx >> 6;
        if (r < nb && col0 + c < nrhs) B[(int64_t)r + (col0 + c) * ldb] = sT[c * LDT + r];
    }
}

int trsm_rec(gpirt_handle_t h, hipStream_t stream, const double* L, int64_t ldl, double* B,
             int64_t nrhs, int64_t ldb, bool trans, int64_t r0, int64_t r1)
{
    const int64_t len = r1 - r0;
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
    // split at a multiple of 64 nearest the middle
    int64_t half = ((len / 2 + NL - 1) / NL) * NL;
    if (half >= len) half = len - NL > 0 ? ((len - 1) / NL) * NL : len / 2;
    const int64_t mid = r0 + half;
    if (!trans) {
        GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, r0, mid));
        // B[mid:r1, :] -= L[mid:r1, r0:mid] * B[r0:mid, :]
        GP_TRY(launch_gemm(h, stream, false, false, TRI_NONE, r1 - mid, nrhs, mid - r0, -1.0,
                           L + mid + r0 * ldl, ldl, B + r0, ldb, 1.0, B + mid, ldb));
        GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, mid, r1));
    } else {
        GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, mid, r1));
        // B[r0:mid, :] -= L[mid:r1, r0:mid]^T * B[mid:r1, :]
        GP_TRY(launch_gemm(h, stream, true, false, TRI_NONE, mid - r0, nrhs, r1 - mid, -1.0,
                           L + mid + r0 * ldl, ldl, B + mid, ldb, 1.0, B + r0, ldb));
        GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, r0, mid));
    }
    return 0;
}

}  // namespace

int launch_trsm_lower(gpirt_handle_t h, hipStream_t stream, const double* L, int64_t n, int64_t ldl,
                      double* B, int64_t nrhs, int64_t ldb, bool trans)
{
    if (n <= 0 || nrhs <= 0) return 0;
    GP_TRY(trsm_rec(h, stream, L, ldl, B, nrhs, ldb, trans, 0, n));
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
