// potrf.hip -- lower Cholesky of S = K + jitter*I : arma::chol(S,"lower") -> LAPACK dpotrf('L')
// (src/gpirtMCMC.cpp:17,78,97), as a blocked right-looking factorisation:
//
//   outer panels of NBO = 1024 columns: trailing update  A22 -= P P^T  (lower blocks only) is an
//       fp64-MFMA syrk with K = 1024 (gemm_f64.hip, 128 x 128 tiles) -- the bulk of the n^3/3 flops;
//   an outer panel is factored in sub-panels of NBP = 512 columns, each by ONE persistent kernel
//       (panel.hip: left-looking per 64-row block, hand-offs through progress counters), with an MFMA
//       update of the outer panel's remaining columns in between (K = 512).
//   Measured at n = 8192 (NBO x NBP swept over 256..2048 x 256..512): flat within 3 % around 1024 x 512.
//
// The launch-per-step panel this replaced (potf2_64 / panel_trsm_64 / K = 64 update gemm, ~3 launches per
// 64 columns, 51 us per step against 26 us now) is still here behind GPIRT_PANEL=2: the reference the persistent
// kernel is tested against, and the path a factorisation is repeated on when the persistent kernel's hang guard
// expired (sampler.hip, guard fallback).
// Switches (Config, common.h) are read once per process / handle, never per call.  The schedules that measured slower
// in rounds 2 and 3 (windowed chain / lean rows kernel / pre-launched chain kernels / row-split and per-panel deferred
// launches / held-back updates) are gone from the tree; DESIGN.md section 4 keeps their numbers.
// Nothing above the diagonal is ever written; the strict upper triangle keeps whatever it held
// (zeros in the sampler's persistent L buffer; the operator entry zero-fills it to honour
// arma::chol's contract).  A non-positive pivot records LAPACK's info (1-based order of the
// leading minor) in h->d_info and lets NaNs propagate; the host checks it after the stream drains.
#include "common.h"

#include <vector>
#include "kernels.h"
#include "solve64.h"
#include "potf2.h"

namespace gpirt {

namespace {

constexpr int NBI = 64;
constexpr int NBO = 1024;
constexpr int NBP = 512;      // widest panel handed to the persistent kernel in one piece

// ------------------------------------------------------------------ diagonal block ---------
__global__ __launch_bounds__(256) void potf2_64_kernel(double* __restrict__ A, int64_t lda, int nb,
                                                       int k0, int* __restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double sP[2 * 4 * NBI];
    __shared__ int sfail;
    potf2_64_body(A, lda, nb, k0, info, sP, &sfail);
}

// ------------------------------------------------------------------ panel solve ------------
// X * Lkk^T = Apanel.  Lkk: nb x nb lower at A[k0,k0]; panel rows r0..n-1, columns k0..k0+nb-1.
// Each wavefront solves 16 rows with the MFMA-layout substitution core (solve64.h); a work-group
// of 4 waves covers 64 rows, so even the last panels keep >= 1 wave per 16 rows in flight.
__global__ __launch_bounds__(256) void panel_trsm_64_kernel(double* __restrict__ A, int64_t lda,
                                                            int64_t n, int64_t k0, int nb, int64_t r0)
{
    __shared__ __attribute__((aligned(16))) double sM[NBI * S64_LS];
    __builtin_amdgcn_s_setprio(3);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int64_t row = r0 + (int64_t)blockIdx.x * 64 + wave * 16 + i;
    const bool live = row < n;
    // the wave's own rows first: these loads do not depend on the staged block and fly while it lands
    d4 X[4];
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * J + 4 * g + r;
            X[J][r] = (live && c < nb) ? A[row + (k0 + c) * lda] : 0.0;
        }
    {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {                // 16 loads in flight, then 16 LDS stores
            const int idx = t + 256 * k;
            const int r = idx & (NBI - 1), c = idx >> 6;
            double x = 0.0;
            if (r < nb && c < nb && r >= c) x = A[(k0 + r) + (k0 + c) * lda];
            else if (r == c) x = 1.0;
            v[k] = x;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = t + 256 * k;
            sM[(idx >> 6) * S64_LS + (idx & (NBI - 1))] = v[k];
        }
    }
    __syncthreads();
    solve64_lower(X, sM);
    if (live) {
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * J + 4 * g + r;
                if (c < nb) A[row + (k0 + c) * lda] = X[J][r];
            }
    }
}

__global__ void zero_upper_kernel(double* __restrict__ A, int64_t n, int64_t lda)
{
    const int64_t c = blockIdx.y;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < c && r < n) A[r + c * lda] = 0.0;
}

}  // namespace

namespace {

// event pair around one syrk launch of the factorisation (only while gpirt_prof_enable is on)
int prof_begin(gpirt_handle_t h, hipStream_t stream, ProfPair& pp) { return prof_pair_begin(h, stream, pp); }
int prof_end(gpirt_handle_t h, hipStream_t stream, ProfPair& pp, int cls, int64_t M, int64_t N, int64_t K)
{
    // algorithmic flops of the lower trapezoid: 2 K (M N - N (N - 1) / 2); bytes: C read + written, P (M x K; its first N
    // rows are the B operand) once
    const double trap = (double)M * (double)N - 0.5 * (double)N * (double)(N - 1);
    return prof_pair_end(h, stream, pp, cls, 2.0 * (double)K * trap, 8.0 * (2.0 * trap + (double)M * (double)K));
}
// C[M x N lower trapezoid] -= P P^T inside an outer panel (between its sub-panels), profiled as class 2
int panel_update(gpirt_handle_t h, hipStream_t stream, int64_t M, int64_t N, int64_t K, const double* P, int64_t ldp,
                 double* C, int64_t ldc)
{
    ProfPair pp;
    GP_TRY(prof_begin(h, stream, pp));
    GP_TRY(launch_gemm(h, stream, false, true, TRI_SYRK_LOWER, M, N, K, -1.0, P, ldp, P, ldp, 1.0, C, ldc));
    return prof_end(h, stream, pp, 2, M, N, K);
}

inline int64_t round64(int v, int dflt) { const int64_t r = ((int64_t)v / NBI) * NBI; return r > 0 ? r : dflt; }

// inner loop of one outer panel: columns [K0, c1), every row below; 64-column steps
// part: 0 = the whole panel; 1 = its first sub-panel only; 2 = the rest (the update of the remaining columns by the first
// sub-panel, then the other sub-panels) -- the halves a distributing host sends one at a time (persistent kernel only)
int factor_panel(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t K0, int64_t c1,
                 bool first_diag_done = false, int part = 0)
{
    if (h->cfg.panel != 2) {
        // the persistent kernel is chain-bound up to ~512 columns and GEMM-bound beyond (one work-group per
        // row block does all of that block's left-looking products): wider outer panels are cut into
        // sub-panels of NBP columns with an MFMA update of the remaining columns in between
        const int64_t nbp = h->cur_nbp;
        for (int64_t k0 = K0; k0 < c1; k0 += nbp) {
            const int64_t k1 = (k0 + nbp < c1) ? k0 + nbp : c1;
            if (!(part == 2 && k0 == K0)) GP_TRY(launch_panel_ll(h, stream, A, n, lda, k0, k1));
            if (part == 1) return 0;
#ifndef GPIRT_EXP_SKIP_INPANEL  // (timing experiment, DESIGN_HISTORY.md section 10: a WRONG factor without this product -- what would absorbing it buy?)
            if (k1 < c1)      // A[k1:n, k1:c1] -= A[k1:n, k0:k1] A[k1:c1, k0:k1]^T
                GP_TRY(panel_update(h, stream, n - k1, c1 - k1, k1 - k0, A + k1 + k0 * lda, lda, A + k1 + k1 * lda, lda));
#endif
        }
        return 0;
    }
    // launch-per-step panel.  part 1 / 2 (the halves of a distributing host): the first sub-panel's columns / the rest --
    // the same launches as the whole panel, cut at the sub-panel boundary
    const int64_t nbp = h->cur_nbp;
    const int64_t mid = (K0 + nbp < c1) ? K0 + nbp : c1;
    const int64_t kb = (part == 2) ? mid : K0, ke = (part == 1) ? mid : c1;
    for (int64_t k0 = kb; k0 < ke; k0 += NBI) {
        const int nb = (int)((c1 - k0) < NBI ? (c1 - k0) : NBI);
        // only the first diagonal block of an outer panel needs its own potf2 launch: every later one is
        // factored by work-group 0 of the previous step's panel update (fused epilogue, gemm_f64.hip)
        if (k0 == K0 && !first_diag_done)
            hipLaunchKernelGGL(potf2_64_kernel, dim3(1), dim3(256), 0, stream, A + k0 + k0 * lda, lda, nb, (int)k0,
                               h->d_info);
        const int64_t r0 = k0 + nb;
        if (r0 >= n) break;
        const int64_t rows = n - r0;
        hipLaunchKernelGGL(panel_trsm_64_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, stream, A, lda,
                           n, k0, nb, r0);
        if (r0 < c1) {
            // rest of the outer panel: A[r0:n, r0:c1] -= A[r0:n, k0:r0] A[r0:c1, k0:r0]^T
            const int nb_next = (int)((c1 - r0) < NBI ? (c1 - r0) : NBI);
            GP_TRY(launch_gemm_update_potf2(stream, rows, c1 - r0, nb, A + r0 + k0 * lda, lda, A + r0 + r0 * lda,
                                            lda, nb_next, (int)r0, h->d_info));
        }
    }
    return 0;
}

// trailing update restricted to the column range [lo, hi):
//   A[lo:n, lo:hi] -= A[lo:n, K0:c1] A[lo:hi, K0:c1]^T      (lower trapezoid, fp64 MFMA syrk)
int trailing(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t K0, int64_t c1,
             int64_t lo, int64_t hi, bool* fused_potf2 = nullptr, bool background = false)
{
    const int64_t M = n - lo, N = hi - lo, K = c1 - K0;
    ProfPair pp;
    GP_TRY(prof_begin(h, stream, pp));
    if (fused_potf2) *fused_potf2 = false;
    if (fused_potf2 && h->cfg.panel == 2 && !gemm_trailing_uses_128(M, N)) {
        // 64-tile launch: work-group 0 also factors the first diagonal block of the next panel
        const int nb_next = (int)(N < NBI ? N : NBI);
        GP_TRY(launch_gemm_update_potf2(stream, M, N, K, A + lo + K0 * lda, lda, A + lo + lo * lda, lda, nb_next,
                                        (int)lo, h->d_info));
        *fused_potf2 = true;
        return prof_end(h, stream, pp, 1, M, N, K);
    }
    GP_TRY(launch_gemm(h, stream, false, true, background ? TRI_SYRK_LOWER_BACKGROUND : TRI_SYRK_LOWER_TRAILING, M, N, K,
                       -1.0, A + lo + K0 * lda, lda, A + lo + K0 * lda, lda, 1.0, A + lo + lo * lda, lda));
    return prof_end(h, stream, pp, gemm_trailing_uses_128(M, N, background) ? 0 : 1, M, N, K);
}


// Panel [K0, c1)'s update of the NEXT panel's block column [lo, hi) -- the update the pivot chain waits for.  Its first
// sub-panel's columns [lo, lo + nbp) receive the panel as TWO products of depth nbp (the panel's first sub-panel, then
// the rest) instead of one of depth c1 - K0: the first half does not need the panel's second sub-panel, so the look-ahead
// schedule applies it while that sub-panel is still being factored and only K = 512 of the update is
// left on the chain.  The rule is the same wherever this block column is updated (with or without look-ahead, in the
// distributed pieces), so L does not depend on the schedule.  Columns [lo + nbp, hi) take the panel in one product.
//   part (bit mask): 1 = the panel's first half on the first sub-panel's columns, 2 = its second half on them (what the
//   chain waits for), 4 = the whole panel on the other columns.  A panel with a single sub-panel has no halves: bit 2
//   then carries the whole product and bit 1 nothing.
int crit_update(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t K0, int64_t c1, int64_t lo,
                int64_t hi, int64_t nbp, int part)
{
    const int64_t a_hi = (lo + nbp < hi) ? lo + nbp : hi;          // the next panel's first sub-panel
    const int64_t kmid = K0 + nbp;                                 // end of this panel's first sub-panel
    if (kmid < c1) {
        if (part & 1) GP_TRY(trailing(h, stream, A, n, lda, K0, kmid, lo, a_hi));
        if (part & 2) GP_TRY(trailing(h, stream, A, n, lda, kmid, c1, lo, a_hi));
    } else if (part & 2) {
        GP_TRY(trailing(h, stream, A, n, lda, K0, c1, lo, a_hi));
    }
    if ((part & 4) && a_hi < hi) GP_TRY(trailing(h, stream, A, n, lda, K0, c1, a_hi, hi));
    return 0;
}

}  // namespace

// ---- the factorisation in pieces, for a host that distributes it (gpirt_amd/distributed.py) ----------
// Outer panel p = columns [p W, min((p + 1) W, n)), W = NBO.  1-D block-cyclic ownership: the owner of panel p factors
// it once its columns carry the updates of every panel q < p, the finished panel travels to the other ranks, and each
// rank applies it to the block columns it owns.  The pieces are exactly the launches of launch_potrf_lower -- same
// sub-panel split, same K = 1024 products applied to every block in ascending panel order -- so the assembled L is
// bit-identical to the single-GPU factor.
// The first sub-panel of an outer panel.  GPIRT_NBP fixes it; otherwise by size: 512 of the 1024 columns, but 704 from
// n = 6144 to 10240 -- measured round 5 on whole iterations at n = 8192 (all three draw_fstar forms, m = 512 ... 2048):
// 6.46 -> 6.38 ms with 576 ... 768 (the factorisation ALONE is unchanged at 4.64 ms: what gains is the sampler's iteration,
// where the side work beside the last outer panel meets a shorter second sub-panel); 2 % slower at 4096, 2.5 - 3 % at
// 12288 and 16384.  One rule for launch_potrf_lower and for the pieces of the distributing hosts (same n, same widths).
int64_t potrf_subpanel_width(int64_t n)
{
    const int set = env_config().nbp;
    if (set > 0) return round64(set, NBP);
    return (n >= 6144 && n <= 10240) ? 704 : NBP;
}
int64_t potrf_panel_width() { return round64(env_config().nbo, NBO); }

// half: 0 = the panel's first sub-panel, 1 = the rest of it, 2 = the whole panel
int potrf_panel_factor(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t p, int64_t extra_rows,
                       int half)
{
    const int64_t W = potrf_panel_width(), K0 = p * W;
    if (p < 0 || K0 >= n || half < 0 || half > 2) { set_error("panel %lld / half %d out of range", (long long)p, half); return GPIRT_E_ARG; }
    h->cur_nbp = potrf_subpanel_width(n);
    return factor_panel(h, stream, A, n + extra_rows, lda, K0, (K0 + W < n) ? K0 + W : n, false, half == 2 ? 0 : half + 1);
}

// A[cW:n, cW:(c+1)W] -= A[cW:n, pW:(p+1)W] A[cW:(c+1)W, pW:(p+1)W]^T   (lower trapezoid of block column c > p)
// part: 2 = everything; 0 = what needs only panel p's FIRST sub-panel (its half of the update of the next panel's first
// columns; nothing for other block columns); 1 = the rest.  0 then 1 launches exactly what 2 does, in the same order.
int potrf_panel_update(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t p, int64_t c,
                       int64_t extra_rows, int part)
{
    const int64_t W = potrf_panel_width(), K0 = p * W, lo = c * W;
    if (p < 0 || c <= p || lo >= n) { set_error("panel update (%lld -> %lld) out of range", (long long)p, (long long)c); return GPIRT_E_ARG; }
    const int64_t hi = (lo + W < n) ? lo + W : n;
    if (part < 0 || part > 2) { set_error("panel update part %d", part); return GPIRT_E_ARG; }
    if (c == p + 1 && h->cfg.panel != 2)        // the next panel's block column: the same products as launch_potrf_lower
        return crit_update(h, stream, A, n + extra_rows, lda, K0, K0 + W, lo, hi, potrf_subpanel_width(n), part == 2 ? 7 : (part == 0 ? 1 : 6));
    if (part == 0) return 0;
    return trailing(h, stream, A, n + extra_rows, lda, K0, K0 + W, lo, hi);
}

// Look-ahead schedule: the trailing update of panel p is split into the columns of panel p+1
// (done first, on the main stream) and the rest; panel p+1 is then factored on a high-priority
// side stream WHILE the rest of update p runs on the main stream.  (A panel work-group holds a whole CU -- 148 KB of
// LDS -- so update work-groups run on the CUs the panel kernel leaves.)
// extra_rows > 0: the matrix is (n + extra_rows) x n -- rows below the square part ride along as ordinary row blocks of
// every panel and trailing update and end up holding  E L^-T  for the rows E they held on entry (a bordered Cholesky:
// draw_fstar's L^-1 K(theta, c) comes out of the factorisation instead of a triangular solve).  Needs n % 64 == 0.
int launch_potrf_lower(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda,
                       bool zero_upper, bool reset_info, int64_t extra_rows)
{
    if (n <= 0) return 0;
    if (extra_rows > 0 && (n % NBI) != 0) { set_error("bordered factorisation needs n %% 64 == 0"); return GPIRT_E_ARG; }
    const int64_t nr = n + extra_rows;          // rows of every panel / update; column limits stay n
    const int64_t nbo = potrf_panel_width();
    const int64_t nbp_la = h->cur_nbp = potrf_subpanel_width(n);
    const bool persistent = h->cfg.panel != 2;
    if (reset_info) GP_HIP(hipMemsetAsync(h->d_info, 0, sizeof(int), stream));
    h->prelast_cols = 0;
    const bool la = (h->cfg.lookahead == 1) && (n > 2 * nbo);
    if (la && !h->side) {
        int lo_pri = 0, hi_pri = 0;
        GP_HIP(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
        GP_HIP(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi_pri));
        GP_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_mid, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_half, hipEventDisableTiming));
    }
    // Deferred trailing updates (GPIRT_DEFER=3, the default up to n = 14336; =2 is the plain right-looking order).
    // Panel q's update of a block column r >= q + 2 is not needed before panel r is factored.  Applying it to the whole
    // rest of the matrix at once (right-looking) loads the main stream with 38.7 GFLOP behind panel 0 and 1.1 behind panel
    // 5, while the side chain needs the same ~0.55 ms every time.  With =3 each step brings ONE more block column up to
    // date: 11.8 / 19.3 / 22.6 / 21.5 / 16.1 / 6.4 GFLOP per step at n = 8192.  Every block of C still receives its
    // rank-1024 updates in ascending panel order, so L is bit-identical (tests/test_gpu_fences.py; the factorisation in
    // pieces of the distributed hosts is tested against it bit for bit).  From ~14000 rows on the plain order's large
    // 128-tile updates win (n = 16384: 28.3 against 29.6 ms, n = 20000: 54.7 against 59.9).
    const int defer = (h->cfg.defer == 2 || h->cfg.defer == 3) ? h->cfg.defer : (n <= 14336 ? 3 : 2);
    std::vector<int64_t> done_col;                   // (mode 3) columns < done_col[q] carry panel q's update
    bool half_done = false;                          // the next crit_update's first half is already in
    GP_TRY(factor_panel(h, stream, A, nr, lda, 0, nbo < n ? nbo : n));
    for (int64_t K0 = 0; K0 < n; K0 += nbo) {
        const int64_t c1 = (K0 + nbo < n) ? K0 + nbo : n;
        if (c1 >= n) break;
        const int64_t c2 = (c1 + nbo < n) ? c1 + nbo : n;
        bool diag_done = false;
        // with persistent sub-panels and look-ahead, only the first sub-panel's columns gate the side stream
        const int64_t cA = (c1 + nbp_la < c2) ? c1 + nbp_la : c2;
        const bool split = la && c2 < n && persistent && cA < c2;
        if (persistent) {
            // (the first sub-panel's columns first: part 1; the others follow behind the fork when `split`)
            GP_TRY(crit_update(h, stream, A, nr, lda, K0, c1, c1, c2, nbp_la, (half_done ? 2 : 3) | (split ? 0 : 4)));
            half_done = false;
        } else {
            GP_TRY(trailing(h, stream, A, nr, lda, K0, c1, c1, c2, &diag_done));
        }
        if (la && c2 < n) {
            GP_HIP(hipEventRecord(h->ev_fork, stream));
            GP_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
            if (split) {
                // the side stream starts on the first sub-panel as soon as ITS columns are up to date; the
                // other columns of the outer panel are brought up to date behind it on the main stream
                GP_TRY(factor_panel(h, h->side, A, nr, lda, c1, cA));
                GP_HIP(hipEventRecord(h->ev_half, h->side));
                GP_TRY(crit_update(h, stream, A, nr, lda, K0, c1, c1, c2, nbp_la, 4));
                GP_HIP(hipEventRecord(h->ev_mid, stream));
                GP_HIP(hipStreamWaitEvent(h->side, h->ev_mid, 0));
                GP_TRY(panel_update(h, h->side, nr - cA, c2 - cA, cA - c1, A + cA + c1 * lda, lda, A + cA + cA * lda, lda));
                GP_TRY(factor_panel(h, h->side, A, nr, lda, cA, c2));
            } else {
                GP_TRY(factor_panel(h, h->side, A, nr, lda, c1, c2, diag_done)); // next panel, side stream
            }
            if (c2 + nbo >= n) {
                // [0, c2) is final and only the last outer panel is left -- a phase with most of the chip idle: whoever has
                // work that needs only the finished part of L (the sampler's block inverses) may start behind this event
                if (!h->ev_prelast) GP_HIP(hipEventCreateWithFlags(&h->ev_prelast, hipEventDisableTiming));
                GP_HIP(hipEventRecord(h->ev_prelast, h->side));
                h->prelast_cols = c2;
            }
            if (defer == 3 && split) {
                done_col.push_back(c2);                                   // this panel: [c1, c2) done above
                const int64_t horizon = (c2 + nbo < n) ? c2 + nbo : n;    // ONE block column brought up to date per step
                // The step's updates of this block column by the panels 0 .. q as ONE grid -- their products side by
                // side, applied to C one after the other in panel order (launch_syrk_panels: bit-identical to separate
                // launches).  Every panel so far has brought the block column up to the same column (done_col), which is
                // what makes them one product over contiguous K.  7.29 -> 7.22 ms per iteration at 8192 x 1024,
                // 12.72 -> 12.18 ms for the factorisation at 12288; the parts' round trip through the workspace takes the
                // launches' memory traffic from 1.9x to 2.35x their algorithmic bytes.
                bool same_lo = done_col.size() >= 2;
                for (size_t q = 1; q < done_col.size(); ++q) same_lo = same_lo && done_col[q] == done_col[0];
                if (same_lo && done_col[0] < horizon && (nbo % 16) == 0) {
                    const int64_t lo = done_col[0], M = nr - lo, N = horizon - lo;
                    const int np = (int)done_col.size();
                    const size_t need = (size_t)np * (size_t)M * (size_t)N * sizeof(double);
                    if (h->defer_ws_bytes < need) {
                        GP_HIP(hipStreamSynchronize(stream));
                        GP_HIP(hipStreamSynchronize(h->side));
                        if (h->d_defer_ws) GP_HIP(hipFree(h->d_defer_ws));
                        h->d_defer_ws = nullptr; h->defer_ws_bytes = 0;
                        const size_t want = need + need / 4;
                        GP_HIP(hipMalloc(&h->d_defer_ws, want));
                        h->defer_ws_bytes = want;
                    }
                    ProfPair pp;
                    GP_TRY(prof_begin(h, stream, pp));
                    GP_TRY(launch_syrk_panels(stream, M, N, nbo, np, -1.0, A + lo, lda, A + lo, lda, A + lo + lo * lda, lda, h->d_defer_ws));
                    GP_TRY(prof_end(h, stream, pp, 1, M, N, (int64_t)np * nbo));      // (one pair: the np products + their application)
                    for (size_t q = 0; q < done_col.size(); ++q) done_col[q] = horizon;
                }
                for (size_t q = 0; q < done_col.size(); ++q)
                    if (done_col[q] < horizon) {
                        GP_TRY(trailing(h, stream, A, nr, lda, (int64_t)q * nbo, (int64_t)(q + 1) * nbo, done_col[q], horizon, nullptr, true));
                        done_col[q] = horizon;
                    }
            } else {
                GP_TRY(trailing(h, stream, A, nr, lda, K0, c1, c2, n));    // the rest, concurrently
            }
            if (split && cA - c1 == nbp_la) {
                // the first sub-panel of the panel being factored on the side stream is final: its half of the NEXT
                // step's chain-critical update goes out now, behind this step's other updates of that block column
                // (ascending panel order), beside the second sub-panel's kernel (4.93 -> 4.82 ms, round 3)
                const int64_t a_hi = (c2 + nbp_la < n) ? c2 + nbp_la : n;
                GP_HIP(hipStreamWaitEvent(stream, h->ev_half, 0));
                GP_TRY(trailing(h, stream, A, nr, lda, c1, cA, c2, a_hi, nullptr, true));
                half_done = true;
            }
            GP_HIP(hipEventRecord(h->ev_join, h->side));
            GP_HIP(hipStreamWaitEvent(stream, h->ev_join, 0));
        } else {
            if (c2 < n) GP_TRY(trailing(h, stream, A, nr, lda, K0, c1, c2, n));
            GP_TRY(factor_panel(h, stream, A, nr, lda, c1, c2, diag_done));
        }
    }
    if (zero_upper) {
        hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256),
                           0, stream, A, n, lda);
    }
    GP_HIP(hipGetLastError());
    h->factor_count += 1;
    if (h->trip_guard_at >= 0 && h->factor_count == h->trip_guard_at) {
        // gpirt_debug_trip_guard: leave behind what a hang-guard expiry leaves -- the guard word raised and a result that
        // was never finished (eight columns of NaN from the middle down) -- without spinning any kernel to its bound
        const int64_t c = n / 2;
        GP_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_info + 1), 1, 1, stream));
        GP_HIP(hipMemset2DAsync(A + c + c * lda, (size_t)lda * 8, 0xFF, (size_t)(n - c) * 8, (size_t)(n - c < 8 ? n - c : 8), stream));
    }
    return 0;
}

// After a hang-guard expiry (info[1] raised): drain the handle's streams, clear the guard record, the potrf info word and
// the progress counters, so that a repeated factorisation starts from a clean slate.  The caller rebuilds the matrix and
// factors again with the launch-per-step panel (h->cfg.panel = 2 around the call).
int potrf_guard_reset(gpirt_handle_t h, hipStream_t stream)
{
    GP_HIP(hipStreamSynchronize(stream));
    if (h->side) GP_HIP(hipStreamSynchronize(h->side));
    if (h->cfg.guard_verbose) {
        int w[8] = {};
        GP_HIP(hipMemcpy(w, h->d_info, sizeof(w), hipMemcpyDeviceToHost));
        fprintf(stderr, "[gpirt] hang-guard fallback: info %d guard %d | %d %d | need %d %d | seen %d %d\n", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
    }
    GP_HIP(hipMemsetAsync(h->d_info, 0, 8 * sizeof(int), stream));
    if (h->d_prog) GP_HIP(hipMemsetAsync(h->d_prog, 0, 2 * h->prog_cap * sizeof(unsigned long long), stream));
    GP_HIP(hipStreamSynchronize(stream));
    h->guard_fallbacks += 1;
    return 0;
}

}  // namespace gpirt
