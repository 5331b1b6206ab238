// solve64.h -- the 64-column triangular substitution core shared by the Cholesky panel solve
// (X Lkk^T = A, src/gpirtMCMC.cpp:17) and the trsm leaves of draw_fstar (src/draw-fstar.cpp:7,19).
//
// One wavefront solves  M x = b  for 16 independent 64-vectors at a time, M lower triangular
// (64 x 64, in LDS).  A vector is kept in the accumulator layout of v_mfma_f64_16x16x4_f64:
// lane (i = lane & 15, g = lane >> 4) holds, for each 16-column block J, the four entries
// 16J + 4g + r (r = 0..3) of vector i.  With that layout
//   * the off-diagonal work  x_J -= M[J,I] x_I  (I < J) is 4 MFMAs per block pair whose B operand
//     is literally the registers that hold x_I -- no shuffle, no LDS round trip for x;
//   * only the 16 x 16 diagonal blocks are solved by substitution: 16 steps, each one division,
//     one cross-lane broadcast (ds_bpermute from the owning 16-lane group) and 4 FMAs per lane,
//     with the block's coefficients pre-loaded into registers so no step waits on LDS.
// True substitution throughout (no inverted blocks): S = K + 1e-3 I has condition ~1e6 here.
#pragma once

#include "common.h"

namespace gpirt {

constexpr int S64_LS = 80;     // LDS column stride (doubles): 64 + 16 keeps both read patterns conflict-free

// sM[c * S64_LS + r] = M[r][c] for r >= c (lower triangle incl. diagonal), zeros above.
__device__ __forceinline__ void solve64_lower(d4 (&X)[4], const double* __restrict__ sM)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const int pi = 4 * (i & 3) + (i >> 2);          // MFMA row index i <-> physical column pi
#pragma unroll
    for (int J = 0; J < 4; ++J) {
        // ---- strip update from the solved blocks I < J :  X_J -= M[J,I] X_I
#pragma unroll
        for (int I = 0; I < J; ++I) {
            double a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = -sM[(16 * I + 4 * g + s) * S64_LS + 16 * J + pi];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                X[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], X[I][s], X[J], 0, 0, 0);
        }
        // ---- diagonal block: coefficients of this lane's four columns, all 16 steps, up front
        double Ld[16][4];
        double dj[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double* col = sM + (16 * J + j) * S64_LS + 16 * J;
            dj[j] = col[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) Ld[j][r] = col[4 * g + r];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int gj = j >> 2, rj = j & 3;
            const double t = X[J][rj] / dj[j];                       // valid on the owner group
            const int src = (i | (gj << 4)) << 2;                    // byte address for bpermute
            const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(t));
            const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(t));
            const double xj = __hiloint2double(hi, lo);
#pragma unroll
            for (int r = 0; r < 4; ++r) X[J][r] = fma(-xj, Ld[j][r], X[J][r]);
            if (g == gj) X[J][rj] = xj;
        }
    }
}

}  // namespace gpirt
