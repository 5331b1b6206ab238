// solve64.h -- the 64-column triangular substitution core shared by the Cholesky panel solve
// (X Lkk^T = A, src/gpirtMCMC.cpp:17) and the trsm leaves of draw_fstar (src/draw-fstar.cpp:7,19).
//
// One wavefront solves  M x = b  for 16 independent 64-vectors at a time, M lower triangular
// (64 x 64, in LDS).  A vector is kept in the accumulator layout of v_mfma_f64_16x16x4_f64:
// lane (i = lane & 15, g = lane >> 4) holds, for each 16-column block J, the four entries
// 16J + 4g + r (r = 0..3) of vector i.  With that layout
//   * the off-diagonal work  x_J -= M[J,I] x_I  (I < J) is 4 MFMAs per block pair whose B operand
//     is literally the registers that hold x_I -- no shuffle, no LDS round trip for x;
//   * only the 16 x 16 diagonal blocks are solved by substitution, in four group steps: a lane
//     group owns four consecutive columns, so each 4 x 4 diagonal sub-block is solved inside the
//     lane, broadcast once (ds_bpermute from the owning 16-lane group) and folded into the other
//     columns with 16 FMAs per lane; coefficients and pivot reciprocals are in registers before
//     the dependent chain starts, so no step waits on LDS or on a division.
// solve64_lower is a true substitution throughout (kept for the launch-per-step Cholesky panel);
// solve64_lower_inv below applies the 16 x 16 diagonal blocks through their inverses and is what the
// persistent panel kernel and the trsm leaves run.  S = K + 1e-3 I has condition ~1e6 here.
#pragma once

#include "common.h"

namespace gpirt {

constexpr int S64_LS = 68;     // LDS column stride (doubles): 4 columns apart = 16 (mod 32) 8-byte banks; 34 KB per block

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
        // ---- 16 x 16 diagonal block in four group steps.  Lane group gq owns columns 4gq..4gq+3 of
        // every vector, so the 4 x 4 diagonal sub-block is solved inside the lane (no cross-lane
        // traffic), the four results are broadcast once from the owner group (ds_bpermute) and
        // every lane folds them into its own four columns.  Reciprocals of the pivots are formed
        // up front (off the dependent chain); coefficients come from LDS before the chain starts.
        double C[4][4][4];                         // C[gq][r][q] = M[16J+4g+r][16J+4gq+q]
        double rinv[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* col = sM + (16 * J + 4 * gq + q) * S64_LS + 16 * J + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) C[gq][r][q] = col[r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) rinv[r] = 1.0 / sM[(16 * J + 4 * g + r) * S64_LS + 16 * J + 4 * g + r];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            // in-lane 4 x 4 forward substitution with this lane's OWN diagonal sub-block
            // (meaningful on the owner group g == gq; the other groups' results are discarded)
            double x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = X[J][r];
#pragma unroll
                for (int q = 0; q < r; ++q) acc = fma(-C[gq][r][q], x[q], acc);
                x[r] = acc * rinv[r];
            }
            const int src = (i | (gq << 4)) << 2;
            double xb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(x[r]));
                const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(x[r]));
                xb[r] = __hiloint2double(hi, lo);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = X[J][r];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = fma(-C[gq][r][q], xb[q], acc);   // zero above the diagonal
                X[J][r] = (g == gq) ? xb[r] : acc;
            }
        }
    }
}

// ---- the same solve with the 16 x 16 diagonal blocks applied as explicit inverses ----------------------
// solve64_lower spends ~3/4 of its time in the 16 group steps of the diagonal blocks (in-lane 4 x 4
// substitution + ds_bpermute, a dependent chain the MFMA pipe only watches).  Inverting the four 16 x 16
// diagonal blocks once per staged matrix (one wavefront, ~0.6 us) turns them into 4 MFMAs each:
//     X_J = (T_J - sum_{I<J} X_I M_JI^T) W_J^T,   W_J = M_JJ^{-1}
// 40 MFMAs per wavefront in all (1.1 us against 4.3 us measured for the substitution).  Only 16 x 16
// blocks of the Cholesky factor are ever inverted: their condition is bounded by that of L itself
// (sqrt(cond S) ~ 1e3 here), far below anything fp64 would notice; the off-diagonal coupling stays a true
// substitution.

// Replace the four diagonal 16 x 16 blocks of the staged lower-triangular matrix by their inverses (zeros
// above the diagonal).  All 256 threads call it after the barrier that completes the staging; it ends with
// a barrier.  Wavefront 0 works: lane = 16 * block + column j of W, rows i = 0..15 in sequence.
__device__ __forceinline__ void invert_diag16(double* __restrict__ sM)
{
    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) == 0) {
        const int blk = lane >> 4, j = lane & 15;
        double* base = sM + (16 * blk) * S64_LS + 16 * blk;     // element (r, c) of the block: base[c * S64_LS + r]
        // forward substitution on the unit vector e_j, column by column of M (right-looking): the updates
        // of one step are independent of each other, and column k+1 of M is fetched (16-byte LDS reads,
        // broadcast within the block's 16 lanes) while step k computes -- no LDS latency on the chain
        double w[16], r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double d = base[i * S64_LS + i];
            double x = __builtin_amdgcn_rcp(d);                  // 1/d: hardware estimate + two Newton steps
            x = fma(fma(-d, x, 1.0), x, x);
            r[i] = fma(fma(-d, x, 1.0), x, x);
            w[i] = (i == j) ? 1.0 : 0.0;
        }
        double mc[16], mn[16];
#pragma unroll
        for (int i2 = 0; i2 < 16; i2 += 2) {
            const double2 v = *reinterpret_cast<const double2*>(&base[0 * S64_LS + i2]);
            mc[i2] = v.x;
            mc[i2 + 1] = v.y;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k + 1 < 16) {
#pragma unroll
                for (int i2 = 0; i2 < 16; i2 += 2)
                    if (i2 + 1 > k + 1) {
                        const double2 v = *reinterpret_cast<const double2*>(&base[(k + 1) * S64_LS + i2]);
                        mn[i2] = v.x;
                        mn[i2 + 1] = v.y;
                    }
            }
            w[k] *= r[k];
#pragma unroll
            for (int i = k + 1; i < 16; ++i) w[i] = fma(-mc[i], w[k], w[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mc[i] = mn[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) base[j * S64_LS + i] = w[i];
    }
    __syncthreads();
}

// sM as for solve64_lower, but with the diagonal 16 x 16 blocks already replaced by invert_diag16.
__device__ __forceinline__ void solve64_lower_inv(d4 (&X)[4], const double* __restrict__ sM)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const int pi = 4 * (i & 3) + (i >> 2);
#pragma unroll
    for (int J = 0; J < 4; ++J) {
#pragma unroll
        for (int I = 0; I < J; ++I) {
            double a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = -sM[(16 * I + 4 * g + s) * S64_LS + 16 * J + pi];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                X[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], X[I][s], X[J], 0, 0, 0);
        }
        double a[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = sM[(16 * J + 4 * g + s) * S64_LS + 16 * J + pi];
        d4 Y = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int s = 0; s < 4; ++s) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], X[J][s], Y, 0, 0, 0);
        X[J] = Y;
    }
}

// X_J -= T X_I for one 64 x 64 coefficient block T (sT[c * S64_LS + r] = T[r][c], full block):
// the off-diagonal step between two 64-row blocks of a taller solve.  64 MFMAs per wavefront.
__device__ __forceinline__ void strip64_update(d4 (&XJ)[4], const d4 (&XI)[4], const double* __restrict__ sT)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const int pi = 4 * (i & 3) + (i >> 2);
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            double a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = -sT[(16 * I + 4 * g + s) * S64_LS + 16 * J + pi];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                XJ[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], XI[I][s], XJ[J], 0, 0, 0);
        }
}

}  // namespace gpirt
