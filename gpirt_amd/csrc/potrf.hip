// potrf.hip -- lower Cholesky of S = K + jitter*I : arma::chol(S,"lower") -> LAPACK dpotrf('L')
// (src/gpirtMCMC.cpp:17,78,97), as a two-level blocked right-looking factorisation:
//
//   outer panels of NBO = 256 columns: trailing update  A22 -= P P^T  (lower blocks only) is one
//       fp64-MFMA syrk launch with K = 256  (gemm_f64.hip, TRI_SYRK_LOWER) -- n^3/3 of the flops;
//   inside a panel, steps of NBI = 64 columns:
//       potf2_64      one work-group factors the 64 x 64 diagonal block in LDS;
//       panel_trsm_64 X L_kk^T = A_panel by substitution, one lane per row (the row lives in 64
//                     fp64 registers, L_kk is broadcast from LDS): rows are independent, so this
//                     is perfectly parallel over the n - k rows below the block;
//       the panel's remaining columns are updated with a K = 64 MFMA gemm (masked to the lower
//       triangle).
// Nothing above the diagonal is ever written; the strict upper triangle keeps whatever it held
// (zeros in the sampler's persistent L buffer; the operator entry zero-fills it to honour
// arma::chol's contract).  A non-positive pivot records LAPACK's info (1-based order of the
// leading minor) in h->d_info and lets NaNs propagate; the host checks it after the stream drains.
#include "common.h"
#include "kernels.h"

namespace gpirt {

namespace {

constexpr int NBI = 64;
constexpr int NBO = 256;
constexpr int LDD = NBI + 1;

// ------------------------------------------------------------------ diagonal block ---------
__global__ __launch_bounds__(256) void potf2_64_kernel(double* __restrict__ A, int64_t lda, int nb,
                                                       int k0, int* __restrict__ info)
{
    __shared__ double sd[NBI * LDD];
    __shared__ int sfail;
    const int t = threadIdx.x;
    if (t == 0) sfail = 0;
    for (int idx = t; idx < NBI * NBI; idx += 256) {
        const int r = idx & (NBI - 1), c = idx >> 6;
        double v = 0.0;
        if (r < nb && c < nb && r >= c) v = A[(int64_t)r + (int64_t)c * lda];
        else if (r == c) v = 1.0;                       // identity padding for nb < 64
        sd[r + c * LDD] = v;
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const double d = sd[j + j * LDD];
        if (!(d > 0.0) && t == 0 && sfail == 0) sfail = j + 1;
        const double piv = sqrt(d);                     // NaN when d < 0: propagates
        __syncthreads();
        if (t > j && t < NBI) sd[t + j * LDD] /= piv;
        if (t == j) sd[j + j * LDD] = piv;
        __syncthreads();
        const int w = nb - j - 1;
        for (int idx = t; idx < w * w; idx += 256) {
            const int r = j + 1 + idx % w, c = j + 1 + idx / w;
            if (r >= c) sd[r + c * LDD] -= sd[r + j * LDD] * sd[c + j * LDD];
        }
        __syncthreads();
    }
    for (int idx = t; idx < NBI * NBI; idx += 256) {
        const int r = idx & (NBI - 1), c = idx >> 6;
        if (r < nb && c < nb && r >= c) A[(int64_t)r + (int64_t)c * lda] = sd[r + c * LDD];
    }
    if (t == 0 && sfail != 0) atomicCAS(info, 0, k0 + sfail);
}

// ------------------------------------------------------------------ panel solve ------------
// X * Lkk^T = Apanel.  Lkk: nb x nb lower at A[k0,k0]; panel rows r0..n-1, columns k0..k0+nb-1.
// Full 64-column panels: one lane per row, the row lives in 64 fp64 registers, Lkk is broadcast
// from LDS.  (Predicating the loads/stores on a run-time nb makes hipcc spill ~3800 registers, so
// the ragged last panel goes through the generic kernel below instead.)
__global__ __launch_bounds__(256) void panel_trsm_64_kernel(double* __restrict__ A, int64_t lda,
                                                            int64_t n, int64_t k0, int64_t r0)
{
    // sLt[c][c2] = L[c2][c]  (column c of Lkk contiguous over c2)
    __shared__ __attribute__((aligned(16))) double sLt[NBI * NBI];
    const int t = threadIdx.x;
    for (int idx = t; idx < NBI * NBI; idx += 256) {
        const int c2 = idx & (NBI - 1), c = idx >> 6;   // element L[c2][c], c2 >= c
        sLt[c * NBI + c2] = (c2 >= c) ? A[(k0 + c2) + (k0 + c) * lda] : 0.0;
    }
    __syncthreads();
    const int64_t r = r0 + (int64_t)blockIdx.x * 256 + t;
    if (r >= n) return;
    double* row = A + r + k0 * lda;
    double x[NBI];
#pragma unroll
    for (int c = 0; c < NBI; ++c) x[c] = row[c * lda];
#pragma unroll
    for (int c = 0; c < NBI; ++c) {
        const double xc = x[c] / sLt[c * NBI + c];
        x[c] = xc;
#pragma unroll
        for (int c2 = c + 1; c2 < NBI; ++c2) x[c2] -= xc * sLt[c * NBI + c2];
    }
#pragma unroll
    for (int c = 0; c < NBI; ++c) row[c * lda] = x[c];
}

// ragged panel (nb < 64): same substitution with the row kept in LDS; 64 rows per work-group
__global__ __launch_bounds__(64) void panel_trsm_ragged_kernel(double* __restrict__ A, int64_t lda,
                                                               int64_t n, int64_t k0, int nb,
                                                               int64_t r0)
{
    __shared__ double sL[NBI * NBI];      // sL[c * 64 + c2] = L[c2][c]
    __shared__ double sx[NBI * 64];       // sx[c * 64 + t]
    const int t = threadIdx.x;
    for (int idx = t; idx < nb * NBI; idx += 64) {
        const int c2 = idx & (NBI - 1), c = idx >> 6;
        sL[c * NBI + c2] = (c2 < nb && c2 >= c) ? A[(k0 + c2) + (k0 + c) * lda] : 0.0;
    }
    __syncthreads();
    const int64_t r = r0 + (int64_t)blockIdx.x * 64 + t;
    if (r >= n) return;
    for (int c = 0; c < nb; ++c) sx[c * 64 + t] = A[r + (k0 + c) * lda];
    for (int c = 0; c < nb; ++c) {
        const double xc = sx[c * 64 + t] / sL[c * NBI + c];
        sx[c * 64 + t] = xc;
        for (int c2 = c + 1; c2 < nb; ++c2) sx[c2 * 64 + t] -= xc * sL[c * NBI + c2];
    }
    for (int c = 0; c < nb; ++c) A[r + (k0 + c) * lda] = sx[c * 64 + t];
}

__global__ void zero_upper_kernel(double* __restrict__ A, int64_t n, int64_t lda)
{
    const int64_t c = blockIdx.y;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < c && r < n) A[r + c * lda] = 0.0;
}

}  // namespace

int launch_potrf_lower(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda,
                       bool zero_upper)
{
    if (n <= 0) return 0;
    GP_HIP(hipMemsetAsync(h->d_info, 0, sizeof(int), stream));
    const bool prof = h->prof.enabled;
    for (int64_t K0 = 0; K0 < n; K0 += NBO) {
        const int64_t c1 = (K0 + NBO < n) ? K0 + NBO : n;      // end of this outer panel
        for (int64_t k0 = K0; k0 < c1; k0 += NBI) {
            const int nb = (int)((c1 - k0) < NBI ? (c1 - k0) : NBI);
            hipLaunchKernelGGL(potf2_64_kernel, dim3(1), dim3(256), 0, stream,
                               A + k0 + k0 * lda, lda, nb, (int)k0, h->d_info);
            const int64_t r0 = k0 + nb;
            if (r0 >= n) break;
            const int64_t rows = n - r0;
            if (nb == NBI)
                hipLaunchKernelGGL(panel_trsm_64_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256),
                                   0, stream, A, lda, n, k0, r0);
            else
                hipLaunchKernelGGL(panel_trsm_ragged_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64),
                                   0, stream, A, lda, n, k0, nb, r0);
            if (r0 < c1) {
                // rest of the outer panel: A[r0:n, r0:c1] -= A[r0:n, k0:r0] A[r0:c1, k0:r0]^T
                GP_TRY(launch_gemm(h, stream, false, true, TRI_SYRK_LOWER, rows, c1 - r0, nb, -1.0,
                                   A + r0 + k0 * lda, lda, A + r0 + k0 * lda, lda, 1.0,
                                   A + r0 + r0 * lda, lda));
            }
        }
        if (c1 < n) {
            const int64_t rows = n - c1;
            ProfPair pp{nullptr, nullptr, 0.0};
            if (prof) {
                if (!h->prof.free_pairs.empty()) { pp = h->prof.free_pairs.back(); h->prof.free_pairs.pop_back(); }
                else { GP_HIP(hipEventCreate(&pp.e0)); GP_HIP(hipEventCreate(&pp.e1)); }
                GP_HIP(hipEventRecord(pp.e0, stream));
            }
            GP_TRY(launch_gemm(h, stream, false, true, TRI_SYRK_LOWER, rows, rows, c1 - K0, -1.0,
                               A + c1 + K0 * lda, lda, A + c1 + K0 * lda, lda, 1.0,
                               A + c1 + c1 * lda, lda));
            if (prof) {
                GP_HIP(hipEventRecord(pp.e1, stream));
                pp.flops = (double)rows * (double)rows * (double)(c1 - K0);   // n^2 k (lower half, 2 flop/fma)
                h->prof.pending.push_back(pp);
            }
        }
    }
    if (zero_upper) {
        hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256),
                           0, stream, A, n, lda);
    }
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
