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
// 64 columns, 51 us per step against 26 us now) is still here behind GPIRT_PANEL=2, as the reference the
// persistent kernel is tested against.
// Nothing above the diagonal is ever written; the strict upper triangle keeps whatever it held
// (zeros in the sampler's persistent L buffer; the operator entry zero-fills it to honour
// arma::chol's contract).  A non-positive pivot records LAPACK's info (1-based order of the
// leading minor) in h->d_info and lets NaNs propagate; the host checks it after the stream drains.
#include "common.h"

#include <vector>
#include "kernels.h"
#include "solve64.h"
#include "potf2.h"

#include <stdlib.h>

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
int prof_begin(gpirt_handle_t h, hipStream_t stream, ProfPair& pp)
{
    pp = ProfPair{nullptr, nullptr, 0.0, 0, 0.0};
    if (!h->prof.enabled) return 0;
    if (!h->prof.free_pairs.empty()) { pp = h->prof.free_pairs.back(); h->prof.free_pairs.pop_back(); }
    else { GP_HIP(hipEventCreate(&pp.e0)); GP_HIP(hipEventCreate(&pp.e1)); }
    GP_HIP(hipEventRecord(pp.e0, stream));
    return 0;
}
int prof_end(gpirt_handle_t h, hipStream_t stream, ProfPair& pp, int cls, int64_t M, int64_t N, int64_t K)
{
    if (!pp.e0) return 0;
    GP_HIP(hipEventRecord(pp.e1, stream));
    // algorithmic flops of the lower trapezoid: 2 K (M N - N (N - 1) / 2)
    const double trap = (double)M * (double)N - 0.5 * (double)N * (double)(N - 1);
    pp.flops = 2.0 * (double)K * trap;
    pp.bytes = 8.0 * (2.0 * trap + (double)M * (double)K);     // C read + written, P (M x K; its first N rows are the B operand) once
    pp.cls = cls;
    h->prof.pending.push_back(pp);
    return 0;
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

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    const int x = atoi(v);
    return x > 0 ? x : dflt;
}

// GPIRT_PANEL=2 selects the launch-per-step panel (potf2 / panel_trsm / update gemm) instead of the
// persistent left-looking kernel of panel.hip
bool panel_persistent()
{
    static const bool on = !(getenv("GPIRT_PANEL") && atoi(getenv("GPIRT_PANEL")) == 2);
    return on;
}

// inner loop of one outer panel: columns [K0, c1), every row below; 64-column steps
// part: 0 = the whole panel; 1 = its first sub-panel only; 2 = the rest (the update of the remaining columns by the first
// sub-panel, then the other sub-panels) -- the halves a distributing host sends one at a time (persistent kernel only)
int factor_panel(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t K0, int64_t c1,
                 bool first_diag_done = false, int part = 0)
{
    static const bool fuse = !(getenv("GPIRT_FUSE_POTF2") && atoi(getenv("GPIRT_FUSE_POTF2")) == 2);
    if (panel_persistent()) {
        // the persistent kernel is chain-bound up to ~512 columns and GEMM-bound beyond (one work-group per
        // row block does all of that block's left-looking products): wider outer panels are cut into
        // sub-panels of NBP columns with an MFMA update of the remaining columns in between
        static const int nbp_env = env_int("GPIRT_NBP", NBP);
        const int64_t nbp = (nbp_env / NBI) * NBI > 0 ? (nbp_env / NBI) * NBI : NBP;
        // GPIRT_ROWS=1: the sub-panel in two launches -- the persistent kernel on the diagonal owners and the `lean_win` rows
        // below them (whole CUs), panel_rows_kernel (32-row work-groups that SHARE CUs with the updates) on the rest,
        // beside it on a helper stream; same counters, same arithmetic, L bit-identical (panel.hip)
        static const int lean_rows = env_int("GPIRT_ROWS", 0);
        static const int lean_win = env_int("GPIRT_ROWS_WINDOW", 512);
        for (int64_t k0 = K0; k0 < c1; k0 += nbp) {
            const int64_t k1 = (k0 + nbp < c1) ? k0 + nbp : c1;
            const int64_t wend = k1 + ((int64_t)lean_win / NBI) * NBI;
            if (lean_rows == 1 && !(part == 2 && k0 == K0) && (k1 - k0) % NBI == 0 && wend + 2048 <= n && (wend % NBI) == 0) {
                if (!h->rows_stream) GP_HIP(hipStreamCreateWithFlags(&h->rows_stream, hipStreamNonBlocking));
                for (int e = 12; e < 14; ++e)
                    if (!h->ev_pool[e]) GP_HIP(hipEventCreateWithFlags(&h->ev_pool[e], hipEventDisableTiming));
                unsigned long long epoch = 0;
                GP_HIP(hipEventRecord(h->ev_pool[12], stream));
                GP_HIP(hipStreamWaitEvent(h->rows_stream, h->ev_pool[12], 0));
                GP_TRY(launch_panel_ll(h, stream, A, n, lda, k0, k1, wend, &epoch));
                GP_TRY(launch_panel_rows(h, h->rows_stream, A, n, lda, k0, k1, wend, n, epoch));
                GP_HIP(hipEventRecord(h->ev_pool[13], h->rows_stream));
                GP_HIP(hipStreamWaitEvent(stream, h->ev_pool[13], 0));
            } else if (!(part == 2 && k0 == K0)) {
                GP_TRY(launch_panel_ll(h, stream, A, n, lda, k0, k1));
            }
            if (part == 1) return 0;
            if (k1 < c1)      // A[k1:n, k1:c1] -= A[k1:n, k0:k1] A[k1:c1, k0:k1]^T
                GP_TRY(panel_update(h, stream, n - k1, c1 - k1, k1 - k0, A + k1 + k0 * lda, lda, A + k1 + k1 * lda, lda));
        }
        return 0;
    }
    if (part != 0) { set_error("the panel in halves needs the persistent panel kernel"); return GPIRT_E_ARG; }
    for (int64_t k0 = K0; k0 < c1; k0 += NBI) {
        const int nb = (int)((c1 - k0) < NBI ? (c1 - k0) : NBI);
        // only the first diagonal block of an outer panel needs its own potf2 launch: every later one is
        // factored by work-group 0 of the previous step's panel update (fused epilogue, gemm_f64.hip)
        if ((k0 == K0 && !first_diag_done) || !fuse)
            hipLaunchKernelGGL(potf2_64_kernel, dim3(1), dim3(256), 0, stream, A + k0 + k0 * lda, lda, nb, (int)k0,
                               h->d_info);
        const int64_t r0 = k0 + nb;
        if (r0 >= n) break;
        const int64_t rows = n - r0;
        hipLaunchKernelGGL(panel_trsm_64_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, stream, A, lda,
                           n, k0, nb, r0);
        if (r0 < c1) {
            // rest of the outer panel: A[r0:n, r0:c1] -= A[r0:n, k0:r0] A[r0:c1, k0:r0]^T
            if (fuse) {
                const int nb_next = (int)((c1 - r0) < NBI ? (c1 - r0) : NBI);
                GP_TRY(launch_gemm_update_potf2(stream, rows, c1 - r0, nb, A + r0 + k0 * lda, lda, A + r0 + r0 * lda,
                                                lda, nb_next, (int)r0, h->d_info));
            } else {
                GP_TRY(launch_gemm(h, stream, false, true, TRI_SYRK_LOWER, rows, c1 - r0, nb, -1.0,
                                   A + r0 + k0 * lda, lda, A + r0 + k0 * lda, lda, 1.0, A + r0 + r0 * lda, lda));
            }
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
    static const bool fuse = !(getenv("GPIRT_FUSE_POTF2") && atoi(getenv("GPIRT_FUSE_POTF2")) == 2);
    if (fused_potf2) *fused_potf2 = false;
    if (fused_potf2 && fuse && !panel_persistent() && !gemm_trailing_uses_128(M, N)) {
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
// schedule applies it while that sub-panel is still being factored (first_half_done) and only K = 512 of the update is
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


// ---- the windowed schedule (round 3, GPIRT_SCHED=2) ------------------------------------------------------------------
// What the pivot chain needs next is never more than the rows of the outer panel being factored and of the one behind
// it; everything further down only has to be ready one outer panel later.  So a sub-panel is no longer ONE kernel over
// all rows, gated by an update of all rows (the schedule below: ~100-170 us of full-chip update + a 165-280 us kernel,
// 16 times in a row).  Per outer panel p = columns [c1, c2), sub-panels A = [c1, cA) and B = [cA, c2), five streams:
//   chain A (high prio)  chainA: panel_ll_kernel on the rows [c1, c2) only (16 work-groups); midW: the K = 512 update of
//                        B's diagonal triangle (split over K: a lone tile per CU runs its K loop at latency); flag B.
//   chain B (high prio)  chainB (8 work-groups).  Both chain launches are PRE launches: enqueued one step ahead, resident
//                        on their compute units while they wait for their input flag -- a panel work-group needs a WHOLE
//                        CU and, launched when its input is ready, waits for the updates beside it to drain (measured:
//                        165 us kernels stretched to 375-515 us).
//   near (high prio)     the rows [c2, c3) of the NEXT outer panel: panel_rows_kernel beside chainA / chainB (lean 32-row
//                        work-groups consuming the chain launches' counters), the K = 512 update between them; then updW
//                        -- panel p's K = 1024 update of the next panel's 1024 x 1024 diagonal trapezoid (split over K) --
//                        and flag A; then updN: panel p's update of the rows [c3, c4) of block column p + 1.
//   rows                 the same for the rest of the rows [c3, nr), then updR (rows below c4 of block column p + 1).
//   main                 the deferred trailing updates (block column p + 2 <- panels 0 .. p, as under GPIRT_DEFER=3).
// Every element of the matrix still receives the same products in the same order -- K = 512 inside an outer panel,
// K = 1024 per earlier panel in ascending order, the left-looking sums of the panel kernels -- except the two split-K
// regions, whose parts are added in a fixed order.  Deadlock freedom: a launch that may spin on counters (near / rows) is
// enqueued after the chain launch it consumes and never fills the chip (launch_panel_rows); a PRE launch waits for a flag
// whose producer sits on ANOTHER hardware queue (probed once per handle, win_setup; otherwise the schedule is not used)
// and depends only on work enqueued before it or on streams that never wait for the chain launches.
namespace {

struct WinStreams { hipStream_t chainA, chainB, near, rows, main; };

int win_setup(gpirt_handle_t h)
{
    if (h->win_state != 0) return 0;
    int lo_pri = 0, hi_pri = 0;
    GP_HIP(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
    if (!h->side) {
        GP_HIP(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi_pri));
        GP_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_mid, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_a, hipEventDisableTiming));
    }
    GP_HIP(hipStreamCreateWithPriority(&h->chainb_stream, hipStreamNonBlocking, hi_pri));
    GP_HIP(hipStreamCreateWithPriority(&h->near_stream, hipStreamNonBlocking, hi_pri));
    GP_HIP(hipStreamCreateWithFlags(&h->rows_stream, hipStreamNonBlocking));
    for (auto& e : h->ev_pool) GP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    GP_HIP(hipMalloc(&h->d_ready, 4 * sizeof(unsigned long long)));
    GP_HIP(hipMalloc(&h->d_chain_ws, (size_t)4 * 1024 * 1024 * sizeof(double)));
    GP_HIP(hipMemsetAsync(h->d_ready, 0, 4 * sizeof(unsigned long long), h->side));
    GP_HIP(hipStreamSynchronize(h->side));
    // the PRE launches (GPIRT_WIN_PRE=1) spin on chain A / chain B for flags raised from chain A / near, and what raises
    // them waits for events of rows / main: a spinner must not sit in front of any of those in a shared hardware queue
    int ok = 1, r = 0;
    if (!(getenv("GPIRT_WIN_PRE") && atoi(getenv("GPIRT_WIN_PRE")) == 1)) { h->win_state = 1; return 0; }
    int* d_res = reinterpret_cast<int*>(h->d_ready + 2);
    hipStream_t spin[2] = { h->side, h->chainb_stream };
    hipStream_t others[4] = { h->near_stream, h->rows_stream, h->stream, nullptr };
    for (int a = 0; a < 2 && ok; ++a) {
        others[3] = spin[1 - a];
        for (int b = 0; b < 4 && ok; ++b) {
            GP_TRY(panel_queue_probe(spin[a], others[b], h->d_ready, d_res, &r));
            ok = ok && r;
        }
    }
    h->win_state = ok ? 1 : 2;
    return 0;
}

// C[rows r0..r1, cols lo..hi] -= A[rows, K0..c1] A[lo..hi, K0..c1]^T : the lower trapezoid when the row range starts at lo
// (syrk mode), a full rectangle when it lies below the column range.  inpanel: the K = 512 update inside an outer panel
// (profiling class 2), otherwise a K = 1024 trailing update (class 0 / 1 by tile).  splitk: the small trapezoids on the
// pivot chain (launch_syrk_splitk).
int win_update(gpirt_handle_t h, hipStream_t st, double* A, int64_t lda, int64_t K0, int64_t c1, int64_t r0, int64_t r1,
               int64_t lo, int64_t hi, bool inpanel, bool splitk = false)
{
    const int64_t M = r1 - r0, N = hi - lo, K = c1 - K0;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    ProfPair pp;
    GP_TRY(prof_begin(h, st, pp));
    const double* P = A + r0 + K0 * lda;
    const double* Q = A + lo + K0 * lda;
    if (r0 == lo) {
        static const int nsplit = env_int("GPIRT_CHAIN_SPLITK", 4);
        if (splitk && nsplit > 1 && M <= 1024 && N <= 1024 && K >= 256) {
            GP_TRY(launch_syrk_splitk(st, M, N, K, -1.0, P, lda, Q, lda, 1.0, A + r0 + lo * lda, lda, h->d_chain_ws, nsplit > 4 ? 4 : nsplit));
            return prof_end(h, st, pp, inpanel ? 2 : 1, M, N, K);
        }
        GP_TRY(launch_gemm(h, st, false, true, inpanel ? TRI_SYRK_LOWER : TRI_SYRK_LOWER_BACKGROUND, M, N, K, -1.0, P, lda, Q, lda,
                           1.0, A + r0 + lo * lda, lda));
        return prof_end(h, st, pp, inpanel ? 2 : (gemm_trailing_uses_128(M, N, true) ? 0 : 1), M, N, K);
    }
    if (r0 < hi) { set_error("windowed update: row range cuts the column range"); return GPIRT_E_ARG; }
    GP_TRY(launch_gemm_nosplit(st, false, true, M, N, K, -1.0, P, lda, Q, lda, 1.0, A + r0 + lo * lda, lda));
    if (!pp.e0) return 0;
    GP_HIP(hipEventRecord(pp.e1, st));
    pp.flops = 2.0 * (double)M * (double)N * (double)K;
    pp.bytes = 8.0 * (2.0 * (double)M * (double)N + (double)(M + N) * (double)K);
    pp.cls = inpanel ? 2 : (gemm_rect_uses_128(M, N) ? 0 : 1);
    h->prof.pending.push_back(pp);
    return 0;
}

int potrf_windowed(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t nr, int64_t nbo, int64_t nbp)
{
    WinStreams S{ h->side, h->chainb_stream, h->near_stream, h->rows_stream, stream };
    hipEvent_t* E = h->ev_pool;
    enum { E_CA = 0, E_CB = 2, E_NB = 4, E_RB = 6, E_DEF = 8, E_MW = 10, E_UW = 11 };   // [+ parity of p]
    unsigned long long* readyA = h->d_ready;
    unsigned long long* readyB = h->d_ready + 1;
    // PRE launches are OFF unless GPIRT_WIN_PRE=1: they measured slower still (6.0 against 5.6 ms) and, once in a full test
    // session (many streams created and destroyed by then), a pre-launched chain kernel's bounded wait for its flag
    // expired -- the hang guard reported it instead of hanging -- although the queue probe had passed for this handle.
    // Without them every spinning launch only waits for work enqueued BEFORE it: no dependence on the queue mapping.
    static const bool pre = (getenv("GPIRT_WIN_PRE") && atoi(getenv("GPIRT_WIN_PRE")) == 1);
    GP_HIP(hipEventRecord(h->ev_fork, stream));
    GP_HIP(hipStreamWaitEvent(S.chainA, h->ev_fork, 0));
    GP_HIP(hipStreamWaitEvent(S.chainB, h->ev_fork, 0));
    GP_HIP(hipStreamWaitEvent(S.near, h->ev_fork, 0));
    GP_HIP(hipStreamWaitEvent(S.rows, h->ev_fork, 0));
    const int64_t P = (n + nbo - 1) / nbo;
    std::vector<int64_t> done_col((size_t)P, 0);      // columns < done_col[q] carry panel q's update (deferred updates)
    unsigned long long eA = 0, eB = 0, eA_next = 0;
    // chain launch of the first sub-panel: its input is the matrix itself
    GP_TRY(launch_panel_ll(h, S.chainA, A, nr, lda, 0, nbp < n ? nbp : n, nbo < nr ? (nbo < n ? nbo : n) : 0, &eA_next, nullptr));
    for (int64_t p = 0; p < P; ++p) {
        const int b = (int)(p & 1), bp = b ^ 1;
        const int64_t c1 = p * nbo, c2 = (c1 + nbo < n) ? c1 + nbo : n;
        const int64_t cA = (c1 + nbp < c2) ? c1 + nbp : c2;
        const int64_t c3 = (c2 < n) ? ((c2 + nbo < n) ? c2 + nbo : n) : c2;     // end of the next outer panel's rows
        const int64_t c4 = (c3 < n) ? ((c3 + nbo < n) ? c3 + nbo : n) : c3;
        const bool has_b = cA < c2;
        const bool last = c2 >= n;
        eA = eA_next;                                   // chainA(p) is already enqueued (above, or by the previous step)
        // ---- chain A: [chainA(p) enqueued earlier] -> E_CA -> midW -> flag B ;  chain B: chainB(p), PRE
        GP_HIP(hipEventRecord(E[E_CA + b], S.chainA));
        if (has_b) {
            if (pre) {
                GP_TRY(launch_panel_ll(h, S.chainB, A, nr, lda, cA, c2, c2 < nr ? c2 : 0, &eB, readyB));
                GP_TRY(win_update(h, S.chainA, A, lda, c1, cA, cA, c2, cA, c2, true, true));
                GP_TRY(launch_flag_store(S.chainA, readyB, eB));
            } else {                                     // (GPIRT_WIN_PRE=0: ordinary launches ordered by events)
                GP_TRY(win_update(h, S.chainA, A, lda, c1, cA, cA, c2, cA, c2, true, true));
                GP_HIP(hipEventRecord(E[E_MW], S.chainA));
                GP_HIP(hipStreamWaitEvent(S.chainB, E[E_MW], 0));
                GP_TRY(launch_panel_ll(h, S.chainB, A, nr, lda, cA, c2, c2 < nr ? c2 : 0, &eB, nullptr));
            }
        } else {
            GP_HIP(hipStreamWaitEvent(S.chainB, E[E_CA + b], 0));     // E_CB always implies E_CA
        }
        GP_HIP(hipEventRecord(E[E_CB + b], S.chainB));
        // the NEXT outer panel's first chain launch goes out now (PRE): resident while chainB(p) runs
        const int64_t nx1 = c2, nx2 = c3, nxA = (nx1 + nbp < nx2) ? nx1 + nbp : nx2;
        if (!last && pre) GP_TRY(launch_panel_ll(h, S.chainA, A, nr, lda, nx1, nxA, nx2 < nr ? nx2 : 0, &eA_next, readyA));
        // ---- near: the next outer panel's rows [c2, c3), in step with the chain launches
        if (c3 > c2) {
            GP_TRY(launch_panel_rows(h, S.near, A, nr, lda, c1, cA, c2, c3, eA));
            if (has_b) {
                GP_HIP(hipStreamWaitEvent(S.near, E[E_CA + b], 0));
                GP_TRY(win_update(h, S.near, A, lda, c1, cA, c2, c3, cA, c2, true));
                GP_TRY(launch_panel_rows(h, S.near, A, nr, lda, cA, c2, c2, c3, eB));
            }
        }
        GP_HIP(hipEventRecord(E[E_NB + b], S.near));
        // ---- rows: everything below
        if (nr > c3) {
            GP_TRY(launch_panel_rows(h, S.rows, A, nr, lda, c1, cA, c3, nr, eA));
            if (has_b) {
                GP_HIP(hipStreamWaitEvent(S.rows, E[E_CA + b], 0));
                GP_TRY(win_update(h, S.rows, A, lda, c1, cA, c3, nr, cA, c2, true));
                GP_TRY(launch_panel_rows(h, S.rows, A, nr, lda, cA, c2, c3, nr, eB));
            }
        }
        GP_HIP(hipEventRecord(E[E_RB + b], S.rows));
        if (last) {                                      // last outer panel: nothing left to update
            GP_HIP(hipStreamWaitEvent(stream, E[E_CB + b], 0));
            GP_HIP(hipStreamWaitEvent(stream, E[E_NB + b], 0));
            GP_HIP(hipStreamWaitEvent(stream, E[E_RB + b], 0));
            break;
        }
        // ---- panel p's update of block column p + 1 = [c2, c3), nearest rows first.  The block column already carries
        // the panels q < p (deferred updates of the previous step: E_DEF of the other parity).
        const bool have_def = p >= 1;
        // near: the trapezoid on the diagonal (needs the rows [c2, c3) of panel p: this stream) -> flag A
        if (have_def) GP_HIP(hipStreamWaitEvent(S.near, E[E_DEF + bp], 0));
        GP_TRY(win_update(h, S.near, A, lda, c1, c2, c2, c3, c2, c3, false, true));
        if (pre) {
            GP_TRY(launch_flag_store(S.near, readyA, eA_next));
        } else {
            GP_HIP(hipEventRecord(E[E_UW], S.near));
            GP_HIP(hipStreamWaitEvent(S.chainA, E[E_UW], 0));
            GP_TRY(launch_panel_ll(h, S.chainA, A, nr, lda, nx1, nxA, nx2 < nr ? nx2 : 0, &eA_next, nullptr));
        }
        // near: the rows [c3, c4) -- needs them of panel p (rows stream)
        if (c4 > c3) {
            GP_HIP(hipStreamWaitEvent(S.near, E[E_RB + b], 0));
            GP_TRY(win_update(h, S.near, A, lda, c1, c2, c3, c4, c2, c3, false));
        }
        // rows: the rest -- needs the rows [c2, c3) of panel p (near)
        if (nr > c4) {
            GP_HIP(hipStreamWaitEvent(S.rows, E[E_NB + b], 0));
            if (have_def) GP_HIP(hipStreamWaitEvent(S.rows, E[E_DEF + bp], 0));
            GP_TRY(win_update(h, S.rows, A, lda, c1, c2, c4, nr, c2, c3, false));
        }
        // ---- main: block column p + 2 = [c3, c4) receives the panels 0 .. p in ascending order; the panels q < p need
        // nothing of this step and go first (work for the chip while the chain runs), panel p waits for its rows
        done_col[(size_t)p] = c3;
        if (c4 > c3) {
            for (int64_t q = 0; q <= p; ++q) {
                if (done_col[(size_t)q] >= c4) continue;
                if (q == p) GP_HIP(hipStreamWaitEvent(stream, E[E_RB + b], 0));
                const int64_t lo = done_col[(size_t)q];
                GP_TRY(trailing(h, stream, A, nr, lda, q * nbo, (q + 1) * nbo, lo, c4, nullptr, true));
                done_col[(size_t)q] = c4;
            }
        }
        GP_HIP(hipEventRecord(E[E_DEF + b], stream));
    }
    return 0;
}

}  // namespace

// ---- the factorisation in pieces, for a host that distributes it (gpirt_amd/distributed.py) ----------
// Outer panel p = columns [p W, min((p + 1) W, n)), W = NBO.  1-D block-cyclic ownership: the owner of panel p factors
// it once its columns carry the updates of every panel q < p, the finished panel travels to the other ranks, and each
// rank applies it to the block columns it owns.  The pieces are exactly the launches of launch_potrf_lower -- same
// sub-panel split, same K = 1024 products applied to every block in ascending panel order -- so the assembled L is
// bit-identical to the single-GPU factor.
int64_t potrf_subpanel_width()
{
    const int64_t v = (env_int("GPIRT_NBP", NBP) / NBI) * NBI;
    return v > 0 ? v : NBP;
}

int64_t potrf_panel_width()
{
    static const int nbo_env = env_int("GPIRT_NBO", NBO);
    return (nbo_env / NBI) * NBI > 0 ? (nbo_env / NBI) * NBI : NBO;
}

// half: 0 = the panel's first sub-panel, 1 = the rest of it, 2 = the whole panel
int potrf_panel_factor(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t p, int64_t extra_rows,
                       int half)
{
    const int64_t W = potrf_panel_width(), K0 = p * W;
    if (p < 0 || K0 >= n || half < 0 || half > 2) { set_error("panel %lld / half %d out of range", (long long)p, half); return GPIRT_E_ARG; }
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
    if (c == p + 1 && panel_persistent())       // the next panel's block column: the same products as launch_potrf_lower
        return crit_update(h, stream, A, n + extra_rows, lda, K0, K0 + W, lo, hi, potrf_subpanel_width(), part == 2 ? 7 : (part == 0 ? 1 : 6));
    if (part == 0) return 0;
    return trailing(h, stream, A, n + extra_rows, lda, K0, K0 + W, lo, hi);
}

// Look-ahead schedule: the trailing update of panel p is split into the columns of panel p+1
// (done first, on the main stream) and the rest; panel p+1 is then factored on a high-priority
// side stream WHILE the rest of update p runs on the main stream.  (A panel work-group holds a whole CU -- 512
// registers per lane -- so update work-groups run on the CUs the panel kernel leaves.)
// Measured gain ~6 % of the factorisation: the pivot chain is fp64-VALU latency-bound and slows down
// (clocks, shared FP64 pipes) while the update runs; CU masks (hipExtStreamCreateWithCUMask) and
// single-occupancy GEMM variants were measured and made it worse.
// extra_rows > 0: the matrix is (n + extra_rows) x n -- rows below the square part ride along as ordinary row blocks of
// every panel and trailing update and end up holding  E L^-T  for the rows E they held on entry (a bordered Cholesky:
// draw_fstar's L^-1 K(theta, c) comes out of the factorisation instead of a triangular solve).  Needs n % 64 == 0.
int launch_potrf_lower(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda,
                       bool zero_upper, bool reset_info, int64_t extra_rows)
{
    if (n <= 0) return 0;
    if (extra_rows > 0 && (n % NBI) != 0) { set_error("bordered factorisation needs n %% 64 == 0"); return GPIRT_E_ARG; }
    const int64_t nr = n + extra_rows;          // rows of every panel / update; column limits stay n
    static const int nbo_env = env_int("GPIRT_NBO", NBO);
    static const int lookahead = env_int("GPIRT_LOOKAHEAD", 1);     // 2 = off
    const int64_t nbo = (nbo_env / NBI) * NBI > 0 ? (nbo_env / NBI) * NBI : NBO;
    if (reset_info) GP_HIP(hipMemsetAsync(h->d_info, 0, sizeof(int), stream));
    h->prelast_cols = 0;
    const bool la = (lookahead == 1) && (n > 2 * nbo);
    {
        // GPIRT_SCHED: 2 = the windowed schedule (chain / near / rows / main streams), 1 = one sub-panel kernel over all rows
        const char* sv = getenv("GPIRT_SCHED");
        const int sched = (sv && *sv) ? atoi(sv) : 1;
        const int64_t nbp_w = (env_int("GPIRT_NBP", NBP) / NBI) * NBI > 0 ? (env_int("GPIRT_NBP", NBP) / NBI) * NBI : NBP;
        const bool cols_on = !(getenv("GPIRT_PANEL_COLS") && atoi(getenv("GPIRT_PANEL_COLS")) == 0);
        if (sched == 2 && la && panel_persistent() && cols_on && (nbo % NBI) == 0) GP_TRY(win_setup(h));
        if (sched == 2 && la && panel_persistent() && cols_on && (nbo % NBI) == 0 && h->win_state == 1) {
            GP_TRY(potrf_windowed(h, stream, A, n, lda, nr, nbo, nbp_w));
            if (zero_upper) {
                hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256),
                                   0, stream, A, n, lda);
            }
            GP_HIP(hipGetLastError());
            return 0;
        }
    }
    if (la && !h->side) {
        int lo_pri = 0, hi_pri = 0;
        GP_HIP(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
        GP_HIP(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi_pri));
        GP_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_mid, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_a, hipEventDisableTiming));
    }
    static const int hold_rest = env_int("GPIRT_HOLD_REST", 2);     // 1: always, 2: never (default), 3: only before 128-tile updates
    const int64_t nbp_la = (env_int("GPIRT_NBP", NBP) / NBI) * NBI > 0 ? (env_int("GPIRT_NBP", NBP) / NBI) * NBI : NBP;
    // Deferred trailing updates (GPIRT_DEFER=3, the default since round 2; =2 is the plain right-looking order).
    // Panel q's update of a block column r >= q + 2 is not
    // needed before panel r is factored.  Applying it to the whole rest of the matrix at once (right-looking)
    // loads the main stream with 38.7 GFLOP behind panel 0 and 1.1 behind panel 5, while the side
    // chain needs the same ~0.55 ms every time.  With =3 each step only brings ONE more block column (GPIRT_DEFER_AHEAD)
    // up to date, one launch per finished panel that has not reached it yet: 11.8 / 19.3 / 22.6 / 21.5 / 16.1 /
    // 6.4 GFLOP per step at n = 8192.  Every block of C still receives its rank-1024 updates in ascending panel
    // order, so L is bit-identical (tools/defer_check.py; the factorisation in pieces of the distributed hosts is
    // tested against it bit for bit).  Measured in round 2: 119.8 it/s against 115.4, all syrk launches together at
    // 0.48 of the MFMA peak against 0.45 -- the block-column launches are 64-tile launches (356 128-tiles at most)
    // whose work-groups turn over every ~40 us, so the panel kernel's work-groups (each needs a whole CU) find room
    // sooner than beside 128-tile updates; =1 fuses a step's launches into one product of depth (q + 1) * 1024
    // (one read-modify-write of C, but 1.2 rounds of long tiles: slower).
    // By size unless GPIRT_DEFER says otherwise: from ~14000 rows on the plain order's large 128-tile updates win
    // (n = 16384: 28.3 against 29.6 ms, n = 20000: 54.7 against 59.9; n = 12288: 14.0 against 13.5, n = 8192: 5.3 against 5.0).
    static const int defer_env = env_int("GPIRT_DEFER", 0);  // 3: one launch per panel and block column, 2: off, 1: fused; 0: by size
    const int defer = defer_env ? defer_env : (n <= 14336 ? 3 : 2);
    std::vector<int64_t> done_col;                   // (mode 3) columns < done_col[q] carry panel q's update
    // GPIRT_HALF_AHEAD=2 switches the early half of the chain-critical update off (the products stay the same: the two
    // halves are then applied back to back when the chain needs them)
    static const bool half_ahead = !(getenv("GPIRT_HALF_AHEAD") && atoi(getenv("GPIRT_HALF_AHEAD")) == 2);
    bool half_done = false;                          // the next crit_update's first half is already in
    if (la && !h->ev_half) GP_HIP(hipEventCreateWithFlags(&h->ev_half, hipEventDisableTiming));
    GP_TRY(factor_panel(h, stream, A, nr, lda, 0, nbo < n ? nbo : n));
    for (int64_t K0 = 0; K0 < n; K0 += nbo) {
        const int64_t c1 = (K0 + nbo < n) ? K0 + nbo : n;
        if (c1 >= n) break;
        const int64_t c2 = (c1 + nbo < n) ? c1 + nbo : n;
        bool diag_done = false;
        // with persistent sub-panels and look-ahead, only the first sub-panel's columns gate the side stream
        const int64_t cA = (c1 + nbp_la < c2) ? c1 + nbp_la : c2;
        const bool split = la && c2 < n && panel_persistent() && cA < c2;
        if (panel_persistent()) {
            // (the first sub-panel's columns first: part 1; the others follow behind the fork when `split`)
            GP_TRY(crit_update(h, stream, A, nr, lda, K0, c1, c1, c2, nbp_la, (half_done ? 2 : 3) | (split ? 0 : 4)));
            half_done = false;
        } else {
            GP_TRY(trailing(h, stream, A, nr, lda, K0, c1, c1, split ? cA : c2, split ? nullptr : &diag_done));
        }
        if (la && c2 < n) {
            GP_HIP(hipEventRecord(h->ev_fork, stream));
            GP_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
            if (split) {
                // the side stream starts on the first sub-panel as soon as ITS columns are up to date; the
                // other columns of the outer panel are brought up to date behind it on the main stream
                GP_TRY(factor_panel(h, h->side, A, nr, lda, c1, cA));
                if (half_ahead) GP_HIP(hipEventRecord(h->ev_half, h->side));
                if (panel_persistent()) GP_TRY(crit_update(h, stream, A, nr, lda, K0, c1, c1, c2, nbp_la, 4));
                else GP_TRY(trailing(h, stream, A, nr, lda, K0, c1, cA, c2));
                GP_HIP(hipEventRecord(h->ev_mid, stream));
                GP_HIP(hipStreamWaitEvent(h->side, h->ev_mid, 0));
                GP_TRY(panel_update(h, h->side, nr - cA, c2 - cA, cA - c1, A + cA + c1 * lda, lda, A + cA + cA * lda, lda));
                if (hold_rest == 1 || (hold_rest == 3 && gemm_trailing_uses_128(nr - c2, n - c2))) {
                    // The large (128-tile) updates are released only once the second sub-panel is ready to go as
                    // well: a panel wave holds all 512 registers of its SIMD slice and cannot squeeze in beside resident
                    // update waves, so it has to be dispatched (high-priority stream) before they fill the chip;
                    // released earlier, update and panel kernel stretch each other 1.5-2x.  Measured with the
                    // pipelined GEMM loop: holding (=3) puts the update kernel at 0.62 of the fp64 MFMA peak instead
                    // of 0.54 but the iteration is 3.5 % slower, so the default (=2) releases every update at once;
                    // =1 holds them all.
                    GP_HIP(hipEventRecord(h->ev_a, h->side));
                    GP_HIP(hipStreamWaitEvent(stream, h->ev_a, 0));
                }
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
            if (defer == 1 && split) {
                const int64_t horizon = (c2 + nbo < n) ? c2 + nbo : n;
                GP_TRY(trailing(h, stream, A, nr, lda, 0, c1, c2, horizon, nullptr, true));
            } else if (defer == 3 && split) {
                done_col.push_back(c2);                                   // this panel: [c1, c2) done above
                static const int ahead = env_int("GPIRT_DEFER_AHEAD", 1);  // block columns brought up to date per step
                const int64_t horizon = (c2 + ahead * nbo < n) ? c2 + ahead * nbo : n;
                // GPIRT_DEFER_SPLIT=1 (off by default): each of these launches is cut in two by rows -- the upper slab (with the
                // trapezoid on the diagonal) stays on this stream, the lower one goes to a second stream.  Different rows of a block
                // column are independent, so the two slabs form two chains of launches whose partial last rounds (a
                // launch of 868 64-tiles on 768 slots takes two rounds) fill each other.  Same products per element: L
                // bit-identical (tools/factor_hash.py).  Small and consistent: 7.38 -> 7.35 ms per iteration at the metric
                // size in four alternating A/B runs, 13.08 -> 12.84 ms for the factorisation at n = 12288 -- within noise of
                // nothing, while two launches that run side by side each measure (HIP events, rocprofv3) as long as both
                // together, which makes every per-launch rate in bench.py's roofline read a quarter lower (0.45 -> 0.36)
                // for the same work.  Not worth a misleading profile: a switch.
                static const int defer_split = env_int("GPIRT_DEFER_SPLIT", 2);
                // GPIRT_DEFER_PAR (1 = default): the step's updates of this block column by the panels 0 .. q as ONE grid
                // -- their products side by side, applied to C one after the other in panel order (launch_syrk_panels:
                // bit-identical to the separate launches below, which GPIRT_DEFER_PAR=2 restores).  Every panel so far
                // has brought the block column up to the same column (done_col), which is what makes them one product
                // over contiguous K.  Round 3 measured +2 % on the factorisation at n = 12288 and nothing at the metric
                // size; with the LDS-DMA panel kernel (shorter chain, the main stream no longer has slack) it is
                // 7.29 -> 7.22 ms per iteration at 8192 x 1024 and 12.72 -> 12.18 ms for the factorisation at 12288, and
                // the launches' rate reads 0.53 instead of 0.49 of peak.  The parts' round trip through the workspace
                // takes the launches' memory traffic from 1.9x to 2.35x their algorithmic bytes.
                static const int defer_par = env_int("GPIRT_DEFER_PAR", 1);
                bool same_lo = done_col.size() >= 2;
                for (size_t q = 1; q < done_col.size(); ++q) same_lo = same_lo && done_col[q] == done_col[0];
                if (defer_par == 1 && same_lo && done_col[0] < horizon && (nbo % 16) == 0) {
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
                bool forked = false;
                for (size_t q = 0; q < done_col.size(); ++q)
                    if (done_col[q] < horizon) {
                        const int64_t lo = done_col[q], rows = nr - lo;
                        if (defer_split == 1 && rows >= 3072) {
                            const int64_t mid = lo + ((rows / 2 + NBI - 1) / NBI) * NBI;
                            if (!forked) {
                                if (!h->rows_stream) GP_HIP(hipStreamCreateWithFlags(&h->rows_stream, hipStreamNonBlocking));
                                for (int e = 12; e < 14; ++e)
                                    if (!h->ev_pool[e]) GP_HIP(hipEventCreateWithFlags(&h->ev_pool[e], hipEventDisableTiming));
                                GP_HIP(hipEventRecord(h->ev_pool[12], stream));
                                GP_HIP(hipStreamWaitEvent(h->rows_stream, h->ev_pool[12], 0));
                                forked = true;
                            }
                            GP_TRY(win_update(h, stream, A, lda, (int64_t)q * nbo, (int64_t)(q + 1) * nbo, lo, mid, lo, horizon, false));
                            GP_TRY(win_update(h, h->rows_stream, A, lda, (int64_t)q * nbo, (int64_t)(q + 1) * nbo, mid, nr, lo, horizon, false));
                        } else {
                            if (forked) {      // a launch over all rows behind split ones: the lower slabs first
                                GP_HIP(hipEventRecord(h->ev_pool[13], h->rows_stream));
                                GP_HIP(hipStreamWaitEvent(stream, h->ev_pool[13], 0));
                                forked = false;
                            }
                            GP_TRY(trailing(h, stream, A, nr, lda, (int64_t)q * nbo, (int64_t)(q + 1) * nbo, lo, horizon, nullptr, true));
                        }
                        done_col[q] = horizon;
                    }
                if (forked) {
                    GP_HIP(hipEventRecord(h->ev_pool[13], h->rows_stream));
                    GP_HIP(hipStreamWaitEvent(stream, h->ev_pool[13], 0));
                }
            } else {
                GP_TRY(trailing(h, stream, A, nr, lda, K0, c1, c2, n));    // the rest, concurrently
            }
            if (split && half_ahead && panel_persistent() && cA - c1 == nbp_la && defer != 1) {
                // the first sub-panel of the panel being factored on the side stream is final: its half of the NEXT
                // step's chain-critical update goes out now, behind this step's other updates of that block column
                // (ascending panel order), beside the second sub-panel's kernel
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
    return 0;
}

}  // namespace gpirt
