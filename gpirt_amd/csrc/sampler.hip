// sampler.hip -- device-resident state and the per-iteration driver of gpirtMCMC()
// (src/gpirtMCMC.cpp:5-117): init, draw_f -> draw_fstar -> draw_theta -> draw_beta -> mu, mu_star
// -> K + jitter -> chol, storage of draws, IRF averaging.  One process drives one GPU; a host that
// shards item columns over several processes calls the stage entry points and puts its collective
// (RCCL through torch.distributed) between gpirt_sampler_theta_partial and _theta_finish.
//
// Two RNG contracts:
//   GPIRT_RNG_ITEM     counter-based sub-streams: all m columns of draw_f are ONE trmm + ONE
//                      elliptical-slice launch;
//   GPIRT_RNG_RSTREAM  exact replay of R's global Mersenne-Twister stream.  The stream is generated
//                      on the host one iteration ahead (a fixed, value-independent sequence), copied
//                      to the device, and consumed through a device-resident cursor, because ess()'s
//                      consumption is data dependent (k_j rejections) and sequential over items.
#include "common.h"
#include "kernels.h"
#include "rstream.h"

#include <stdlib.h>
#include <new>
#include <string>
#include <vector>

using namespace gpirt;

namespace {

constexpr int KSPLIT = 16;       // K ranges of the split-K product B^T [B | W] in the low-rank draw_fstar

enum { ST_DRAW_F = 0, ST_FSTAR, ST_THETA_GEMM, ST_THETA_SAMPLE, ST_BETA, ST_FACTOR, ST_COUNT };
const char* const kStageNames[ST_COUNT] = { "draw_f", "draw_fstar", "theta_gemm", "theta_sample",
                                            "draw_beta", "factor" };

}  // namespace

struct gpirt_sampler_s {
    gpirt_handle_t h = nullptr;
    int64_t n = 0, m = 0, N = GPIRT_NGRID;
    gpirt_options opt{};
    RStream* rs = nullptr;            // borrowed (R-stream mode)
    // device state
    double *y = nullptr, *Ypm = nullptr, *theta = nullptr, *theta_new = nullptr, *f = nullptr,
           *Z = nullptr, *NU = nullptr, *beta = nullptr, *mu = nullptr, *mu_star = nullptr,
           *fstar = nullptr, *L = nullptr, *tstar = nullptr, *kstar = nullptr, *rhs = nullptr,
           *mean = nullptr, *s = nullptr, *Gpm = nullptr, *logpost = nullptr, *irf_sum = nullptr,
           *pm = nullptr, *ps = nullptr, *step = nullptr;
    // low-rank K* (opt.kstar_rank = r > 0): Chebyshev nodes, interpolation matrix V (N x r), split-K parts
    int kr = 0;
    double *knodes = nullptr, *kV = nullptr, *kparts = nullptr, *kP = nullptr;
    // L is stored (n + ext) x n with leading dimension ldl.  ext = kr when the low-rank K* is on and n % 64 == 0: the
    // extra rows enter the factorisation holding K(c, theta) (r x n) and leave it holding (L^-1 K(theta, c))^T -- the
    // forward solve of draw_fstar comes out of the bordered factorisation (potrf.hip) for ~1 % more work.
    // Without the low-rank K* (ext_grid): the rows are K(theta*, theta) for the 1001 grid points (padded to 1024 zero
    // rows) and come out as (L^-1 k*)^T -- the 1001-column forward solve of src/draw-fstar.cpp:19 rides through the
    // factorisation the same way.  rows_valid: the rows belong to the current (theta, L); cleared when either is set
    // from outside, rebuilt by an explicit solve on demand (rebuild_rows).
    int64_t ext = 0, ldl = 0;
    bool ext_grid = false, rows_valid = false;
    // Work that needs only L (not this iteration's f) runs on a stream of the sampler's own, beside draw_f's elliptical
    // slice kernel: the part of the low-rank draw_fstar that depends on the factor alone (C = L^-T B, G = B^T B).
    // haux is the main handle's side handle (h->aux, shared by the samplers of that handle, not owned) on that stream (its own trsm / split-K workspaces: several
    // samplers may share `h`).  prep_valid: C and G belong to the current L; *_pending: the main stream has not yet
    // waited for the event.
    gpirt_handle_t haux = nullptr;
    hipEvent_t ev_trmm = nullptr, ev_prep = nullptr;
    bool prep_valid = false, prep_pending = false;
    // the N(0,1) draws of the NEXT draw_f depend on (seed, iteration, item, index) only: filled on the sampler's own stream
    // while the last outer panel is factored; z_filled_iter = the iteration they belong to (0: none), ev_zfill fires when done
    uint32_t z_filled_iter = 0;
    hipEvent_t ev_zfill = nullptr;
    // draw_beta (+ mu, mu_star) needs theta and f but nothing of the factorisation that follows it: it is held back
    // (beta_deferred) until the factorisation has been enqueued and then runs on the sampler's own stream in the same idle
    // phase; beta_pending: the main stream has not yet waited for it (ev_beta).  Anything that reads or writes beta, mu,
    // mu_star, theta or f first flushes / joins it (beta_sync).
    bool beta_deferred = false, beta_pending = false;
    hipEvent_t ev_beta = nullptr;
    // respondent-block form of draw_theta for item-sharded runs (gpirt_sampler_set_theta_block): this rank's
    // block of respondents with ALL items
    int64_t blk_i0 = 0, blk_n = 0, blk_m = 0;
    double *Ypm_blk = nullptr, *Gpm_full = nullptr, *logpost_blk = nullptr, *fstar_full = nullptr, *theta_stage = nullptr;
    // the log-posterior product in fixed point (theta_fixed.hip): byte indicators (built once), digit planes of G, row scales
    TfDims tfd{}, tfd_blk{};
    double *tf_y8 = nullptr, *tf_gq = nullptr, *tf_aux = nullptr, *tf_y8_blk = nullptr, *tf_gq_blk = nullptr, *tf_aux_blk = nullptr;
    int *ess_k = nullptr, *flags = nullptr;    // flags[0] = err, flags[1] = degenerate theta count
    int *fstar_off = nullptr;                  // R-stream replay: consumption offsets of draw_fstar
    int *h_flags = nullptr;                    // pinned
    // R-stream replay
    double* U = nullptr; double* hU = nullptr; uint64_t U_cap = 0;
    uint64_t* pos = nullptr; uint64_t* h_pos = nullptr; uint64_t* beta_off = nullptr;
    uint64_t beta_total = 0;
    RStream saved{};
    bool stream_open = false;
    // The host generator runs AHEAD of the chain (stream_begin / stream_end): while the device works off an iteration the
    // host tops up a FIFO of fresh uniforms (hA, uploaded to S on cs), and the next window is put together on the device
    // from the unconsumed tail of this one + the FIFO's head -- no 65 ms rewind and no 73 ms generation between two
    // iterations (they were half of the default contract's 286 ms at 8192 x 1024).  ahead_gen / ahead_used: draws generated /
    // consumed since the generator was attached; snaps: its state every 2^16 draws, so that the state at the consumed
    // position -- what the caller's RStream must hold whenever anybody looks (rstream_sync) -- is one copy + a short skip.
    gpirt_rstream_s* rs_obj = nullptr;
    bool ahead_attached = false;
    uint64_t ahead_gen = 0, ahead_used = 0, a_len = 0;
    std::vector<std::pair<uint64_t, RStream>> snaps;
    uint32_t *hA = nullptr, *Sraw = nullptr;      // the FIFO as the host produces it: raw Mersenne-Twister words (pinned / device)
    double *S = nullptr, *U2 = nullptr;
    hipStream_t cs = nullptr;
    hipEvent_t ev_up = nullptr, ev_asm = nullptr;
    // draw_f of the replay, three items per pass over L (rng_ess.hip): Nrm = the normal that starts at every position of the
    // window, rs_part = the parts of a pass's 48 candidate products, next_item = the first item no pass has resolved yet
    bool spec_ok = false;
    uint64_t *posv = nullptr, *anchor = nullptr, *h_next = nullptr;
    unsigned long long* rs_flags = nullptr; double* rs_partial = nullptr; uint64_t rs_tag = 0;
    double* Lt = nullptr;             // L in the candidate products' tile order (rebuilt at the start of every draw_f)
    double *Nrm = nullptr, *rs_part = nullptr;
    uint32_t* rs_units = nullptr; int rs_nunits = 0, rs_nfull = 0;
    uint32_t* rs_unitsP = nullptr; int rs_nunitsP = 0, rs_nfullP = 0;     // the predictor's own units (RS3P_KC columns each)
    long long* rs_trace = nullptr;    // debug stamps of one pass (gpirt_debug_rs_trace)
    // the predicted replay (rs_predict.hip): single-precision tiles of L and parts, the predictor's own anchor / cursor / error
    // word, what the verification leaves per item, ctl = [first item not committed, mispredictions, predictor stalls, passes]
    float *Lt32 = nullptr, *rs_part32 = nullptr;
    uint64_t *anchorP = nullptr, *rs_ctl = nullptr, *rs_posP = nullptr;
    double *rs_dec_part = nullptr, *rs_dec_rec = nullptr; unsigned* rs_dec_ticket = nullptr;
    double rs_pass_rate = 0.0; uint64_t rs_pass_seen = 0;    // real predictor passes per item of the last round / the counter's last reading
    int *rs_kpred = nullptr, *rs_kv = nullptr, *rs_used = nullptr, *rs_ierr = nullptr, *rs_errP = nullptr;
    // the predictor's structured form (rs_lr.hip): nodes / weights / K(c, c), the basis at theta in both precisions, prefix Grams,
    // C as tiles, the parts' records, the units; lr_on: in use; lr_strikes: draws in a row whose structured rounds mispredicted
    double *lr_nodes = nullptr, *lr_wts = nullptr, *lr_M = nullptr, *lr_V64 = nullptr, *lr_Gb = nullptr;
    float *lr_V32t = nullptr, *lr_Ct32 = nullptr, *lr_Y = nullptr;
    int* lr_bad = nullptr; uint32_t* lr_units = nullptr; int lr_nunits = 0;
    bool lr_on = false; int lr_strikes = 0; uint64_t rs_mis_seen = 0;
    // bookkeeping
    int iter = 0;                     // completed iterations
    bool initialised = false;
    bool sticky_info = false;         // gpirt_mcmc: potrf info is not cleared between iterations
    bool factor_fresh = false;        // the L enqueued last has not been read by any stage yet (recover_factor)
    bool counted = false;             // registered in h->live_samplers (a fully created sampler)
    bool in_init = false;             // do_factor runs for gpirt_sampler_init (no prefill of Z: init fills it itself)
    bool timing = false;
    hipEvent_t ev[ST_COUNT + 1] = {};
    double stage_ms[ST_COUNT] = {};
    std::vector<void*> allocs;
    std::vector<double> host_tmp;
};

namespace {

template <typename T>
int dalloc(gpirt_sampler_s* s, T** p, size_t count)
{
    void* q = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return GPIRT_E_ALLOC;
    }
    s->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}

inline bool stream_mode(const gpirt_sampler_s* s) { return s->opt.rng_kind == GPIRT_RNG_RSTREAM; }

int report_degenerate_theta(gpirt_sampler_s* s, int count);

// ---- R-stream window ---------------------------------------------------------------------------
// draws [ahead_gen, ahead_gen + count) of the attached generator into dst, a state snapshot every 2^16 draws.  The host
// only runs the Mersenne-Twister recurrence (0.85 ns per word on one core: 16 ms for the 18.8 M draws of an iteration at
// 8192 x 1024); tempering and the conversion to unif_rand()'s double happen on the device (rs_unpack_kernel) -- done on the
// host they were three quarters of the 70 ms this took, more than the device needs for the whole iteration.
void ahead_generate(gpirt_sampler_s* s, uint32_t* dst, uint64_t count)
{
    constexpr uint64_t EVERY = 1ull << 16;
    while (count) {
        const uint64_t c = count < EVERY ? count : EVERY;
        s->snaps.emplace_back(s->ahead_gen, *s->rs);
        s->rs->fill_raw(dst, c);
        s->ahead_gen += c; dst += c; count -= c;
    }
}

// puts the caller's generator back to exactly the consumed position and forgets everything generated ahead
// (gone: the RStream object itself is being destroyed -- the sampler must not touch it again)
void ahead_resolve(void* owner, bool gone)
{
    gpirt_sampler_s* s = static_cast<gpirt_sampler_s*>(owner);
    if (s->ahead_attached) {
        if (s->ev_up) hipEventSynchronize(s->ev_up);          // (an upload may still be reading hA)
        size_t best = 0;
        for (size_t q = 0; q < s->snaps.size(); ++q) if (s->snaps[q].first <= s->ahead_used) best = q;
        *s->rs = s->snaps[best].second;
        s->rs->skip(s->ahead_used - s->snaps[best].first);
    }
    s->ahead_attached = false;
    s->snaps.clear();
    s->ahead_gen = s->ahead_used = s->a_len = 0;
    if (s->rs_obj) { s->rs_obj->owner = nullptr; s->rs_obj->resolve = nullptr; }
    if (gone) { s->rs_obj = nullptr; s->rs = nullptr; }
}

// gpirt_rstream_destroy: the stream object this sampler was created on is being freed
void rstream_forget(void* sampler)
{
    gpirt_sampler_s* s = static_cast<gpirt_sampler_s*>(sampler);
    s->rs_obj = nullptr; s->rs = nullptr;
}

int stream_begin(gpirt_sampler_s* s, uint64_t count)
{
    if (count > s->U_cap) { set_error("R-stream window %llu exceeds capacity %llu", (unsigned long long)count, (unsigned long long)s->U_cap); return GPIRT_E_RNG; }
    if (!s->rs) { set_error("the R stream of this sampler has been destroyed"); return GPIRT_E_ARG; }
    hipStream_t st = s->h->stream;
    if (!s->ahead_attached) {
        // the first window after creation (or after somebody looked at the generator): generate, upload
        if (s->rs_obj->owner && s->rs_obj->owner != s) rstream_sync(s->rs_obj);     // another sampler runs ahead on this stream
        s->snaps.clear();
        s->ahead_gen = s->ahead_used = s->a_len = 0;
        GP_HIP(hipEventSynchronize(s->ev_up));                // (an earlier upload may still be reading hA)
        ahead_generate(s, s->hA, count);
        GP_HIP(hipStreamWaitEvent(st, s->ev_up, 0));          // (... or unpacking Sraw on the copy stream)
        GP_HIP(hipMemcpyAsync(s->Sraw, s->hA, count * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        GP_TRY(launch_rs_unpack(st, s->Sraw, (int64_t)count, s->U));
        GP_HIP(hipEventRecord(s->ev_asm, st));
        GP_HIP(hipEventRecord(s->ev_up, st));                 // (hA and Sraw are free again when this has run)
        s->ahead_attached = true;
        s->rs_obj->owner = s; s->rs_obj->resolve = ahead_resolve;
    }                                                         // (otherwise stream_end has already put the window together)
    GP_HIP(hipMemsetAsync(s->pos, 0, sizeof(uint64_t), st));
    s->stream_open = true;
    return 0;
}

// Fills the FIFO of fresh uniforms up to one window and uploads the new part on the copy stream: host work for the time the
// device is busy.  Called with the items of draw_f enqueued (80 % of an iteration's device time is still ahead), and again
// from stream_end for whoever did not come through there.
int ahead_topup(gpirt_sampler_s* s, uint64_t count)
{
    if (!s->ahead_attached || s->a_len >= count) return 0;
    const uint64_t need = count - s->a_len;
    GP_HIP(hipEventSynchronize(s->ev_up));                    // the previous upload has left hA
    ahead_generate(s, s->hA + s->a_len, need);
    GP_HIP(hipStreamWaitEvent(s->cs, s->ev_asm, 0));          // the last assembly has finished moving S's leftover
    GP_HIP(hipMemcpyAsync(s->Sraw + s->a_len, s->hA + s->a_len, need * sizeof(uint32_t), hipMemcpyHostToDevice, s->cs));
    GP_TRY(launch_rs_unpack(s->cs, s->Sraw + s->a_len, (int64_t)need, s->S + s->a_len));
    GP_HIP(hipEventRecord(s->ev_up, s->cs));
    s->a_len = count;
    return 0;
}

// reads back the cursor, builds the next window on the device
int stream_end(gpirt_sampler_s* s, uint64_t count)
{
    hipStream_t st = s->h->stream;
    GP_HIP(hipMemcpyAsync(s->h_pos, s->pos, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    GP_HIP(hipMemcpyAsync(s->h_flags, s->flags, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    GP_TRY(ahead_topup(s, count));                            // (normally done already, beside draw_f)
    GP_HIP(hipStreamSynchronize(st));
    s->stream_open = false;
    const uint64_t used = *s->h_pos <= count ? *s->h_pos : count;
    s->ahead_used += used;
    if (s->h_flags[0] != 0 || s->h_flags[1] != 0) ahead_resolve(s, false);       // the chain stops here: hand the generator back
    if (s->h_flags[0] != 0) {
        set_error(s->h_flags[0] == GPIRT_E_RNG ? "R-stream replay ran out of pre-generated uniforms"
                  : s->h_flags[0] == GPIRT_E_HIP ? "slice sampler: the work-groups of one item did not meet within the spin bound (device hang guard)"
                  : "sampler state is not finite (elliptical slice sampler met a NaN log-likelihood or did not terminate)");
        return s->h_flags[0];
    }
    // draw_theta as written underflows to 0/0 for some respondents (quirk Q5: the reference then reads theta_star[N] out
    // of bounds): reported at the iteration it happens in, not as whatever the NaN theta breaks next
    if (s->h_flags[1] != 0) return report_degenerate_theta(s, s->h_flags[1]);
    // next window = this one's unconsumed tail + the FIFO's first `used` draws; the FIFO's leftover moves to its front
    // (through the old window buffer: the ranges overlap)
    const uint64_t tail = count - used, rem = s->a_len - used;
    GP_HIP(hipStreamWaitEvent(st, s->ev_up, 0));
    if (tail) GP_HIP(hipMemcpyAsync(s->U2, s->U + used, tail * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (used) GP_HIP(hipMemcpyAsync(s->U2 + tail, s->S, used * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (rem) {
        GP_HIP(hipMemcpyAsync(s->U, s->S + used, rem * sizeof(double), hipMemcpyDeviceToDevice, st));
        GP_HIP(hipMemcpyAsync(s->S, s->U, rem * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    GP_HIP(hipEventRecord(s->ev_asm, st));
    std::swap(s->U, s->U2);
    GP_HIP(hipEventSynchronize(s->ev_up));
    s->a_len = rem;                                           // (the host copy of the FIFO is never read again: nothing to move)
    // snapshots below the consumed position are never needed again (all but the last of them)
    size_t keep = 0;
    for (size_t q = 0; q < s->snaps.size(); ++q) if (s->snaps[q].first <= s->ahead_used) keep = q;
    if (keep) s->snaps.erase(s->snaps.begin(), s->snaps.begin() + (std::ptrdiff_t)keep);
    return 0;
}

uint64_t stream_window(const gpirt_sampler_s* s)
{
    const uint64_t n = (uint64_t)s->n, m = (uint64_t)s->m, N = (uint64_t)s->N;
    return m * (2 * n + 2) + 512 * m + 4096 + 2 * N * m + n + s->beta_total;
}

// nu = L z for ONE right-hand side -- the R-stream replay's rmvnorm (src/mvnormal.h:10).  Under R's stream item j's
// normals start where item j - 1's data-dependent slice loop stopped consuming (src/draw-f.cpp:40-58), so the m products
// of an iteration are DEPENDENT (no batching, and nothing of item j can start beside item j - 1's loop): what is left is
// to make one product cost what its bytes cost.  It is memory-bound -- the lower triangle, n^2 / 2 doubles, read once --
// so the shape is chosen for bytes in flight: 32 rows per work-group (lane = row: 256-byte row segments per column), 8
// column phases per work-group, and TRMV_SPLIT column ranges of every row block side by side (1024 work-groups at n = 8192,
// four per CU), their parts added in a fixed order.  Round 3's thread-per-row kernel (one 64-thread work-group per 64
// rows, each thread walking its whole row) kept 8192 loads in flight on the chip.
// (The strict upper triangle of L holds zeros -- gpirt_sampler_create -- so the diagonal 32 x 32 block is read whole.)
constexpr int TRMV_ROWS = 32, TRMV_SPLIT = 4;

__global__ __launch_bounds__(256) void trmv_lower_part_kernel(const double* __restrict__ L, int64_t n, int64_t ldl,
                                                              const double* __restrict__ z, double* __restrict__ part)
{
    __shared__ double sred[8][TRMV_ROWS];
    const int r = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * TRMV_ROWS, row = row0 + r;
    const int64_t kall = (row0 + TRMV_ROWS < n) ? row0 + TRMV_ROWS : n;        // columns that can be non-zero in these rows
    const int64_t chunk = ((kall + TRMV_SPLIT - 1) / TRMV_SPLIT + 7) / 8 * 8;
    const int64_t k0 = (int64_t)blockIdx.y * chunk, k1 = (k0 + chunk < kall) ? k0 + chunk : kall;
    double acc = 0.0;
    if (row < n) {
        const double* Lr = L + row;
        int64_t k = k0 + ph;
        for (; k + 24 < k1; k += 32) {                    // four loads in flight per lane
            const double a0 = Lr[k * ldl], a1 = Lr[(k + 8) * ldl], a2 = Lr[(k + 16) * ldl], a3 = Lr[(k + 24) * ldl];
            acc += a0 * z[k]; acc += a1 * z[k + 8]; acc += a2 * z[k + 16]; acc += a3 * z[k + 24];
        }
        for (; k < k1; k += 8) acc += Lr[k * ldl] * z[k];
    }
    sred[ph][r] = acc;
    __syncthreads();
    if (ph == 0 && row < n) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += sred[q][r];
        part[(int64_t)blockIdx.y * n + row] = s;
    }
}

__global__ void trmv_sum_kernel(const double* __restrict__ part, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < TRMV_SPLIT; ++q) s += part[(int64_t)q * n + i];
    out[i] = s;
}

int launch_trmv_lower(hipStream_t st, const double* L, int64_t n, int64_t ldl, const double* z, double* part, double* out)
{
    hipLaunchKernelGGL(trmv_lower_part_kernel, dim3((unsigned)((n + TRMV_ROWS - 1) / TRMV_ROWS), TRMV_SPLIT), dim3(256), 0, st,
                       L, n, ldl, z, part);
    hipLaunchKernelGGL(trmv_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, n, out);
    GP_HIP(hipGetLastError());
    return 0;
}

int fstar_prep(gpirt_sampler_s* s, gpirt_handle_t hh);
int report_degenerate_theta(gpirt_sampler_s* s, int count);
int rebuild_rows(gpirt_sampler_s* s);
int beta_sync(gpirt_sampler_s* s);

// Z may still be being filled on the sampler's own stream (do_factor's prefill of the NEXT draw_f's normals): whoever reads or
// rewrites Z, or changes the iteration the prefill was keyed by, joins it first and drops it
int z_sync(gpirt_sampler_s* s)
{
    if (s->z_filled_iter != 0 && s->ev_zfill) GP_HIP(hipStreamWaitEvent(s->h->stream, s->ev_zfill, 0));
    s->z_filled_iter = 0;
    return 0;
}

int do_draw_f(gpirt_sampler_s* s)
{
    gpirt_handle_t h = s->h;
    hipStream_t st = h->stream;
    const int64_t n = s->n, m = s->m;
    const uint32_t iter = (uint32_t)(s->iter + 1);
    GP_TRY(beta_sync(s));                     // mu of the previous iteration's draw_beta
    s->factor_fresh = false;
    if (!stream_mode(s)) {
        const bool prep = s->haux && s->ext > 0 && !s->ext_grid && s->rows_valid && !s->prep_valid;
        // The side work (what the low-rank draw_fstar needs from L alone: the last range of block inverses and the
        // 64-column transposed solve, ~0.6 ms of small DEPENDENT launches) starts at once, beside the fill and the product
        // nu = L z.  Until the end of round 3 it waited for the product with all 1024 columns ("its 256 work-groups would
        // starve it"): what starved was ONE kernel, the 392-register leaf of the block inverses, which cannot share a CU
        // with a 224-register work-group of the product and sat out its whole 1.09 ms -- the side handle now launches
        // the 256-register form of that leaf (trsm.hip, slim_leaf) and the chain is done before the slice kernel is:
        // draw_fstar no longer waits 0.33 ms for it (7.12 -> 6.88 ms per iteration; the product slows by 0.1 ms).
        // GPIRT_PREP_EARLY=2 puts it back behind the product.
        const bool early = prep && !(h->cfg.prep_early == 2 && m > 128);
        if (early) GP_HIP(hipEventRecord(s->ev_trmm, st));
        if (s->z_filled_iter == iter && s->ev_zfill) {
            GP_HIP(hipStreamWaitEvent(st, s->ev_zfill, 0));       // filled behind the previous factorisation (do_factor)
            s->z_filled_iter = 0;
        } else {
            GP_TRY(z_sync(s));                                    // a prefill for ANOTHER iteration may still be writing Z
            GP_TRY(launch_item_uniforms(st, s->opt.seed, iter, GPIRT_ST_F_Z, (uint32_t)s->opt.item0, m, n, s->Z, true));
        }
        {
            ProfPair pp;                                          // (bench.py's roofline: class 3, n^2 m flops)
            GP_TRY(prof_pair_begin(h, st, pp));
            GP_TRY(launch_gemm(h, st, false, false, TRI_A_LOWER, n, m, n, 1.0, s->L, s->ldl, s->Z, n, 0.0, s->NU, n));
            GP_TRY(prof_pair_end(h, st, pp, 3, (double)n * (double)n * (double)m, 8.0 * (0.5 * (double)n * (double)n + 2.0 * (double)n * (double)m)));
        }
        if (prep && !early) GP_HIP(hipEventRecord(s->ev_trmm, st));
        EssArgs a{};
        a.f = s->f; a.nu = s->NU; a.y = s->y; a.mu = s->mu; a.n = n; a.m = m; a.k_out = s->ess_k;
        a.err = s->flags; a.seed = s->opt.seed; a.iter = iter; a.item0 = (uint32_t)s->opt.item0;
        a.ll_exact = h->cfg.ll_exact;
        a.screen = h->cfg.ess_screen == 1;
        GP_TRY(launch_ess(st, a));
        if (prep) {
            // what the low-rank draw_fstar needs from L alone, on the sampler's own stream (see `early` above)
            hipStream_t ax = s->haux->stream;
            GP_HIP(hipStreamWaitEvent(ax, s->ev_trmm, 0));
            s->haux->slim_leaf = early;
            const int rc_prep = fstar_prep(s, s->haux);
            s->haux->slim_leaf = false;
            GP_TRY(rc_prep);
            GP_HIP(hipEventRecord(s->ev_prep, ax));
            s->prep_valid = true; s->prep_pending = true;
        }
        return 0;
    }
    // exact R order: item j draws its n normals, then u, eps0 and one uniform per rejection
    auto plain_item = [&](int64_t j) -> int {
        GP_TRY(launch_rstream_normals(st, s->U, s->pos, 0, n, 1, s->Z));
        GP_TRY(launch_trmv_lower(st, s->L, n, s->ldl, s->Z, s->NU + n, s->NU));        // (the parts go behind NU's first column)
        EssArgs a{};
        a.f = s->f + j * n; a.nu = s->NU; a.y = s->y + j * n; a.mu = s->mu + j * n; a.n = n; a.m = 1;
        a.k_out = s->ess_k + j; a.err = s->flags; a.item0 = (uint32_t)j;
        a.U = s->U; a.pos = s->pos; a.cap = s->U_cap;
        return launch_ess(st, a);
    };
    const bool spec = s->spec_ok;
    if (!spec) {
        for (int64_t j = 0; j < m; ++j) GP_TRY(plain_item(j));
        return 0;
    }
    // Three items per pass over L (rng_ess.hip): a pass is anchored at the first item whose start is known, computes L z for
    // the one place its normals start at and for the 15 + 32 places the next two items' normals can start at, and runs the
    // three slice loops; a slice loop that ran longer than its successors' candidates reach just ends the pass early.  All of
    // it is device-side state (next_item, posv): the host enqueues ceil(m / 3) passes and a few spare ones -- a pass that finds
    // every item done leaves at once -- and reads the item counter back once.
    bool tiles64 = false;                       // (L as fp64 tiles: only the one-phase replay reads them -- built when it first runs)
    Rs3Args a{};
    a.U = s->U; a.cap = s->U_cap; a.Nrm = s->Nrm; a.anchor = s->anchor; a.pos = s->pos; a.posv = s->posv; a.k_out = s->ess_k;
    a.err = s->flags; a.n = n; a.m = m; a.Lt = s->Lt; a.nkb = rs_tile_quads(n); a.part = s->rs_part;
    a.units = s->rs_units; a.nunits = s->rs_nunits; a.nfull = s->rs_nfull;
    a.lim1 = RS3_C1; a.lim2 = RS3_C2;
    if (h->rs_cand_limit > 0) {
        if (h->rs_cand_limit < a.lim1) a.lim1 = h->rs_cand_limit;
        if (h->rs_cand_limit < a.lim2) a.lim2 = h->rs_cand_limit;
    }
    a.f = s->f; a.y = s->y; a.mu = s->mu; a.partial = s->rs_partial; a.flags = s->rs_flags;
    // the normals draw_f can reach: m items of 2n + 2 uniforms + their rejections (the window's own slack, stream_window)
    GP_TRY(launch_rs3_begin(st, a, (uint64_t)m * (2ull * (uint64_t)n + 2ull) + 512ull * (uint64_t)m + 4096ull));
    int64_t pass = 0, done = 0;
    bool topped = false;
    // Predict + verify (rs_predict.hip; GPIRT_RS_PREDICT=2: every pass in fp64, the one-phase replay).  The predictor's
    // passes leave predicted starts; phase B computes every item at its predicted start exactly and commits, in order, what
    // the prediction did not break.  A round that commits nothing (the predictor stalled on its first item) hands the rest
    // of the draw to the one-phase replay, which always makes progress and reports genuine errors.
    bool predict = h->cfg.rs_predict != 2;
    // The structured form of the pass (rs_lr.hip): the blocks of L below the diagonal parts as V C, built from theta alone.  It
    // serves the draw's rounds until one of them finds a misprediction (the dense pass takes the rest of the draw over); three
    // such draws in a row, or a coefficient block the construction itself flags, switch it off for this sampler.
    bool lr = predict && s->lr_on, lr_missed = false;
    // (single-precision tiles of L: the structured pass reads the diagonal parts only; the whole triangle follows if the dense pass is needed)
    bool tiles32_full = !lr;
    if (predict) GP_TRY(launch_rs32_tiles(st, s->L, n, s->ldl, s->Lt32, lr));
    if (lr) {
        RsLrSetup q{};
        q.theta = s->theta; q.n = n; q.nodes = s->lr_nodes; q.wts = s->lr_wts; q.Mn = s->lr_M; q.eps = GPIRT_JITTER; q.L = s->L; q.ldl = s->ldl;
        q.V64 = s->lr_V64; q.Gb = s->lr_Gb; q.V32t = s->lr_V32t; q.Ct32 = s->lr_Ct32; q.nk8 = rs32_tile_octs(n); q.bad = s->lr_bad;
        GP_TRY(launch_rs_lr_setup(st, q));
    }
    while (done < m) {
        const int64_t left = m - done;
        int64_t count = (left + RS3_SLOTS - 1) / RS3_SLOTS + left / 40 + 2;
        if (pass > 4 * m + 64) { set_error("R-stream replay: draw_f made no progress"); return GPIRT_E_NUMERIC; }   // (a pass resolves >= 1 item)
        if (predict) {
            // (a slice loop that rejects 16 points in a row costs a pass of its own.)  How many passes a draw needs is a property
            // of the chain's state and moves slowly: the predictor counts its real passes (rs_ctl[3]) and the next draw enqueues
            // that many per item + 3 % + 4 -- a pass that finds every item predicted leaves at once, but 120 of them per draw
            // were 0.5 ms; too few only costs one more round of both phases for the items left over
            count = (left + RS3_SLOTS - 1) / RS3_SLOTS + left / 8 + 4;
            if (s->rs_pass_rate > 0.0) {
                const int64_t hint = (int64_t)(s->rs_pass_rate * 1.03 * (double)left) + 4;
                if (hint < count) count = hint;
            }
            Rs3Args ap = a;
            ap.anchor = s->anchorP; ap.pos = s->rs_posP; ap.k_out = s->rs_kpred; ap.err = s->rs_errP;
            ap.units = s->rs_unitsP; ap.nunits = s->rs_nunitsP; ap.nfull = s->rs_nfullP;
            if (!lr && !tiles32_full) { GP_TRY(launch_rs32_tiles(st, s->L, n, s->ldl, s->Lt32)); tiles32_full = true; }
            ap.lr = lr ? 1 : 0;
            if (lr) {
                ap.units = s->lr_units; ap.nunits = ap.nfull = s->lr_nunits;
                ap.Ct32 = s->lr_Ct32; ap.lrY = s->lr_Y; ap.V32t = s->lr_V32t;
            }
            ap.Lt32 = s->Lt32; ap.nk8 = rs32_tile_octs(n); ap.part32 = s->rs_part32; ap.mispredict = h->rs_mispredict;
            ap.dec_part = s->rs_dec_part; ap.dec_rec = s->rs_dec_rec; ap.dec_ticket = s->rs_dec_ticket; ap.pass_count = s->rs_ctl + 3;
            GP_TRY(launch_rs_pred_start(st, s->anchor, s->anchorP, s->ess_k, m));
            for (int64_t q = 0; q < count; ++q, ++pass) {
                s->rs_tag += 1ull << 20;
                ap.tag = s->rs_tag;
                ap.trace = (h->rs_trace_pass >= 0 && pass == h->rs_trace_pass) ? s->rs_trace : nullptr;
                ProfPair pp;                                      // (bench.py's roofline: class 4, the lower triangle as floats)
                GP_TRY(prof_pair_begin(h, st, pp));
                GP_TRY(launch_rs3p_products(st, ap));
                // (algorithmic bytes / flops of the pass: the lower triangle as floats -- or, structured, the diagonal parts' rows and C)
                const double pass_el = lr ? (double)(RS3P_KC + RS_LR_RANK) * (double)n : 0.5 * (double)n * (double)(n + 1);
                GP_TRY(prof_pair_end(h, st, pp, 4, 2.0 * RS3_CAND * pass_el, 4.0 * pass_el));
                if (lr) GP_TRY(launch_rs_lr_apply(st, ap));
                GP_TRY(launch_rs3p_decide(st, ap));
            }
            // phase B: the items [done, predicted) at their predicted starts, exactly
            const int64_t mc = m - done;
            GP_TRY(launch_rs_gather(st, s->Nrm, s->posv, s->anchorP, n, done, m, s->Z));
            GP_TRY(launch_gemm(h, st, false, false, TRI_A_LOWER, n, mc, n, 1.0, s->L, s->ldl, s->Z, n, 0.0, s->NU, n));
            RsVerifyArgs v{};
            v.f = s->f; v.nu = s->NU; v.y = s->y; v.mu = s->mu; v.n = n; v.m = m; v.j0 = done;
            v.U = s->U; v.cap = s->U_cap; v.posv = s->posv; v.anchorP = s->anchorP;
            v.kv = s->rs_kv; v.used = s->rs_used; v.ierr = s->rs_ierr;
            GP_TRY(launch_rs_verify(st, v));
            GP_TRY(launch_rs_commit(st, v, s->anchor, s->pos, s->rs_ctl, s->flags, s->f, s->ess_k));
        } else {
        if (!tiles64) { GP_TRY(launch_rs_tiles(st, s->L, n, s->ldl, s->Lt)); tiles64 = true; }
        for (int64_t q = 0; q < count; ++q, ++pass) {
            s->rs_tag += 1ull << 20;
            a.tag = s->rs_tag;
            a.trace = (h->rs_trace_pass >= 0 && pass == h->rs_trace_pass) ? s->rs_trace : nullptr;
            ProfPair pp;                                          // (bench.py's roofline: class 4, the lower triangle's bytes)
            GP_TRY(prof_pair_begin(h, st, pp));
            GP_TRY(launch_rs3_products(st, a));
            GP_TRY(prof_pair_end(h, st, pp, 4, 2.0 * RS3_CAND * 0.5 * (double)n * (double)(n + 1), 8.0 * 0.5 * (double)n * (double)(n + 1)));
            GP_TRY(launch_rs3_slice(st, a));
        }
        }
        GP_HIP(hipMemcpyAsync(s->h_next, s->anchor, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        GP_HIP(hipMemcpyAsync(s->h_next + 1, s->flags, sizeof(int), hipMemcpyDeviceToHost, st));
        if (predict) GP_HIP(hipMemcpyAsync(s->h_next + 2, s->rs_ctl + 3, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        if (predict) GP_HIP(hipMemcpyAsync(s->h_next + 3, s->rs_ctl + 1, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        if (lr) GP_HIP(hipMemcpyAsync(s->h_next + 4, s->lr_bad, sizeof(int), hipMemcpyDeviceToHost, st));
        if (s->stream_open && !topped) { GP_TRY(ahead_topup(s, stream_window(s))); topped = true; }     // the next window's uniforms, while the items run
        GP_HIP(hipStreamSynchronize(st));
        if ((int)s->h_next[1] != 0) break;                    // (an error flag: stream_end / check report it)
        if (predict) {
            if (lr && (int)s->h_next[4] != 0) { s->lr_on = false; lr = false; }                  // the construction flagged itself
            if (lr && s->h_next[3] != s->rs_mis_seen) { lr = false; lr_missed = true; }          // a misprediction: dense from here
            s->rs_mis_seen = s->h_next[3];
        }
        if ((int64_t)s->h_next[0] <= done) {
            if (predict) {                                    // the predictor stalled on its first item: the one-phase replay goes on
                predict = false;
                GP_TRY(launch_advance_pos(st, s->rs_ctl + 2, 1));     // (counted: gpirt_sampler_get "rs_stats"[2])
                continue;
            }
            set_error("R-stream replay: draw_f made no progress"); return GPIRT_E_NUMERIC;
        }
        if (predict && (int64_t)s->h_next[0] > done) {
            // real passes per item of this round (the counter is monotonic: difference to the last reading)
            const uint64_t real = s->h_next[2] - s->rs_pass_seen;
            s->rs_pass_seen = s->h_next[2];
            s->rs_pass_rate = (double)real / (double)((int64_t)s->h_next[0] - done);
        }
        done = (int64_t)s->h_next[0];
    }
    if (s->lr_on) {
        s->lr_strikes = lr_missed ? s->lr_strikes + 1 : 0;
        if (s->lr_strikes >= 3) s->lr_on = false;
    }
    return 0;
}

// The factor-only part of the low-rank draw_fstar (bordered layout): B^T (r x n) sits in the rows below L;
//   Cu = L^-T B  (n x r: transpose, then the transposed solve through the block inverses of L),
//   G  = B^T B   (r x r, split-K over n).
// Runs on hh->stream with hh's workspaces: the sampler's private handle beside draw_f, or the caller's handle inline.
int fstar_prep(gpirt_sampler_s* s, gpirt_handle_t hh)
{
    hipStream_t st = hh->stream;
    const int64_t n = s->n;
    const int r = s->kr;
    double* Cu = s->rhs + (size_t)n * r;
    const double* Bt = s->L + n;
    GP_TRY(launch_transpose(st, Bt, r, n, s->ldl, Cu, n));
    if (hh == s->haux && hh->inv_partial_L == s->L && hh->inv_partial_pairs > 0 && hh->trsm_winv_L != s->L) {
        // most of the block inverses were built while the last outer panel was being factored (do_factor): the rest now.
        // (inv_partial_L lives on the shared side handle: another sampler's early build in between resets it, and this
        // one then rebuilds everything below, in launch_trsm_lower.)
        GP_TRY(trsm_inverses_build(hh, st, s->L, n, s->ldl, true, hh->inv_partial_pairs, (n / 256) / 2));
        trsm_inverses_mark(hh, s->L, n, s->ldl, true);
        hh->inv_partial_L = nullptr; hh->inv_partial_pairs = 0;
    }
    // the block inverses of L are reused when this handle already holds them (built by an earlier solve against the
    // same factor) -- invalidate_factor_products() clears that whenever L changes
    GP_TRY(launch_trsm_lower(hh, st, s->L, n, s->ldl, Cu, r, n, true, true));
    return launch_gemm_splitk(st, false, true, r, r, n, 1.0, Bt, s->ldl, Bt, s->ldl, s->kparts, r, (int64_t)r * r, KSPLIT,
                              s->kP, r, 0.0);
}

int do_draw_fstar(gpirt_sampler_s* s, uint32_t iter)
{
    gpirt_handle_t h = s->h;
    hipStream_t st = h->stream;
    const int64_t n = s->n, m = s->m, N = s->N;
    double* tmp = s->rhs;                    // n x N : L^-1 kstar
    double* W = s->rhs + (size_t)n * N;      // n x m : L^-1 f, then L^-T L^-1 f
    const bool fused = s->opt.fstar_fused != 0;
    GP_TRY(beta_sync(s));                     // mu_star
    s->factor_fresh = false;
    GP_TRY(rebuild_rows(s));
    if (s->kr > 0) {
        // K*^T = U V^T exactly (see gpirt_sampler_create), so with B = L^-1 U and C = L^-T B = S^-1 U:
        //   ||L^-1 k*_j||^2 = v_j^T (B^T B) v_j          (src/draw-fstar.cpp:19-20)
        //   k*_j^T S^-1 f   = v_j^T (C^T f)              (:7, :24-25)
        // Two solves with r right-hand sides each (forward, then transposed with the same block inverses)
        // replace the solve with N + m; the N x n x m product shrinks to r x n x (r + m).
        const int r = s->kr;
        double* Bu = s->rhs;                                  // n x r : U, then B
        double* Cu = s->rhs + (size_t)n * r;                  // n x r : B, then C
        const bool bordered = s->ext > 0;
        if (bordered) {
            // C = L^-T B and G = B^T B depend on the factor alone: usually already under way on the sampler's own
            // stream since draw_f's product finished (do_draw_f); otherwise computed here
            if (!s->prep_valid) { GP_TRY(fstar_prep(s, h)); s->prep_valid = true; }   // (rows rebuilt above if they were stale)
            else if (s->prep_pending) { GP_HIP(hipStreamWaitEvent(st, s->ev_prep, 0)); s->prep_pending = false; }
        } else {
            GP_TRY(launch_se_kernel(st, s->theta, n, s->knodes, r, Bu, n, 0.0));
            GP_TRY(launch_trsm_lower(h, st, s->L, n, s->ldl, Bu, r, n, false));
            GP_HIP(hipMemcpyAsync(Cu, Bu, sizeof(double) * (size_t)n * r, hipMemcpyDeviceToDevice, st));
            GP_TRY(launch_trsm_lower(h, st, s->L, n, s->ldl, Cu, r, n, true, true));
        }
        double* G = s->kP;                                    // r x r
        double* Q = s->kP + (size_t)r * r;                    // r x m
        if (!bordered)
            GP_TRY(launch_gemm_splitk(st, true, false, r, r, n, 1.0, Bu, n, Bu, n, s->kparts, r, (int64_t)r * r, KSPLIT, G, r, 0.0));
        GP_TRY(launch_gemm_splitk(st, true, false, r, m, n, 1.0, Cu, n, s->f, n, s->kparts, r, (int64_t)r * m, KSPLIT, Q, r, 0.0));
        GP_TRY(launch_lowrank_s(st, s->kV, N, r, G, r, s->s));
        GP_TRY(launch_gemm(h, st, false, false, TRI_NONE, N, m, r, 1.0, s->kV, N, Q, r, 0.0, s->mean, N));
        FstarEpiArgs a{};
        a.mean = s->mean; a.mu_star = s->mu_star; a.s = s->s; a.out = s->fstar; a.N = N; a.m = m;
        a.seed = s->opt.seed; a.iter = iter; a.item0 = (uint32_t)s->opt.item0; a.err = s->flags;
        if (stream_mode(s)) { a.U = s->U; a.pos = s->pos; a.cap = s->U_cap; a.off_scratch = s->fstar_off; }
        return launch_fstar_epilogue(st, a);
    }
    GP_HIP(hipMemcpyAsync(W, s->f, sizeof(double) * (size_t)n * m, hipMemcpyDeviceToDevice, st));
    if (s->ext_grid) {
        // tmp = L^-1 k* came out of the bordered factorisation as the rows below L (N x n): one transpose puts it
        // where the solve used to leave it; only the m item columns are still solved for                      :19
        if (!fused) GP_TRY(launch_se_kernel(st, s->theta, n, s->tstar, N, s->kstar, n, 0.0));  // :17 (for :25)
        GP_TRY(launch_transpose(st, s->L + n, N, n, s->ldl, tmp, n));
        GP_TRY(launch_trsm_lower(h, st, s->L, n, s->ldl, W, m, n, false));                    // :7 inner
    } else {
        if (fused) {
            GP_TRY(launch_se_kernel(st, s->theta, n, s->tstar, N, tmp, n, 0.0));               // :17
        } else {
            GP_TRY(launch_se_kernel(st, s->theta, n, s->tstar, N, s->kstar, n, 0.0));
            GP_HIP(hipMemcpyAsync(tmp, s->kstar, sizeof(double) * (size_t)n * N, hipMemcpyDeviceToDevice, st));
        }
        GP_TRY(launch_trsm_lower(h, st, s->L, n, s->ldl, s->rhs, N + m, n, false));                // :19, :7 inner
    }
    GP_TRY(launch_colnorm_s(st, tmp, n, N, n, s->s));                                         // :20
    if (fused) {
        GP_TRY(launch_gemm(h, st, true, false, TRI_NONE, N, m, n, 1.0, tmp, n, W, n, 0.0, s->mean, N));
    } else {
        // (the block inverses of L are the forward solve's: same factor, not modified since)
        GP_TRY(launch_trsm_lower(h, st, s->L, n, s->ldl, W, m, n, true, true));                    // :7 outer
        GP_TRY(launch_gemm(h, st, true, false, TRI_NONE, N, m, n, 1.0, s->kstar, n, W, n, 0.0, s->mean, N)); // :25
    }
    FstarEpiArgs a{};
    a.mean = s->mean; a.mu_star = s->mu_star; a.s = s->s; a.out = s->fstar; a.N = N; a.m = m;
    a.seed = s->opt.seed; a.iter = iter; a.item0 = (uint32_t)s->opt.item0; a.err = s->flags;
    if (stream_mode(s)) { a.U = s->U; a.pos = s->pos; a.cap = s->U_cap; a.off_scratch = s->fstar_off; }
    return launch_fstar_epilogue(st, a);                                                      // :26-28
}

int do_theta_partial(gpirt_sampler_s* s)
{
    hipStream_t st = s->h->stream;
    const int64_t n = s->n, m = s->m, N = s->N;
    const int64_t Np = (N + 127) / 128 * 128;     // Gpm is padded to whole 128-row tiles (zero rows)
    // logpost (N x n) = G+ Y+^T + G- Y-^T   (draw-theta.cpp:15-19 summed over this rank's items): in exact fixed point on the
    // int8 matrix cores (theta_fixed.hip); the fp64 product runs behind it only under the device flag that says a row of G
    // could not be scaled (|f*| beyond exp()'s range, non-finite), or instead of it with GPIRT_THETA_FIXED=2
    const int* only_if = nullptr;
    if (s->h->cfg.theta_fixed == 1) {
        GP_TRY(launch_theta_fixed(st, s->fstar, N, n, m, s->tfd, s->tf_y8, s->tf_gq, s->tf_aux, s->logpost, N, false, s->h));
        only_if = tf_overflow(s->tf_aux, s->tfd);
    }
    GP_TRY(launch_loglik_terms(st, s->fstar, N, m, s->Gpm, Np, only_if));
    return launch_gemm(s->h, st, false, true, TRI_NONE, N, n, 2 * m, 1.0, s->Gpm, Np, s->Ypm, n, 0.0, s->logpost, N, Np, only_if);
}

int do_theta_finish(gpirt_sampler_s* s)
{
    hipStream_t st = s->h->stream;
    ThetaArgs a{};
    a.logpost = s->logpost; a.N = s->N; a.n = s->n; a.stabilise = s->opt.theta_stabilise;
    a.seed = s->opt.seed; a.iter = (uint32_t)(s->iter + 1);
    a.theta_out = s->theta; a.degenerate = s->flags + 1; a.err = s->flags;
    if (stream_mode(s)) { a.U = s->U; a.pos = s->pos; a.cap = s->U_cap; }
    GP_TRY(launch_theta_sample(st, a));
    if (stream_mode(s)) GP_TRY(launch_advance_pos(st, s->pos, (uint64_t)s->n));
    return 0;
}

// draw_theta for the respondents [blk_i0, blk_i0 + blk_n) from the gathered f* of ALL items: the same product and
// the same sampling kernel as the single-GPU path, restricted to a column block of the log-posterior -- so the
// sums over items run in one GEMM in item order (bit-identical to one GPU) and what crosses the links per
// iteration is f* (N x m, 8 MB at the metric size) instead of the N x n partial log-posterior (66 MB).
int do_theta_block(gpirt_sampler_s* s)
{
    hipStream_t st = s->h->stream;
    const int64_t N = s->N, nb = s->blk_n, mt = s->blk_m;
    const int64_t Np = (N + 127) / 128 * 128;
    GP_HIP(hipMemsetAsync(s->theta_stage, 0, sizeof(double) * (size_t)s->n, st));
    if (nb == 0) return 0;
    const int* only_if = nullptr;
    if (s->h->cfg.theta_fixed == 1) {                       // (as do_theta_partial; exact sums: the same bits as one GPU's product)
        GP_TRY(launch_theta_fixed(st, s->fstar_full, N, nb, mt, s->tfd_blk, s->tf_y8_blk, s->tf_gq_blk, s->tf_aux_blk, s->logpost_blk, N, false, s->h));
        only_if = tf_overflow(s->tf_aux_blk, s->tfd_blk);
    }
    GP_TRY(launch_loglik_terms(st, s->fstar_full, N, mt, s->Gpm_full, Np, only_if));
    GP_TRY(launch_gemm(s->h, st, false, true, TRI_NONE, N, nb, 2 * mt, 1.0, s->Gpm_full, Np, s->Ypm_blk, nb, 0.0,
                       s->logpost_blk, N, Np, only_if));
    ThetaArgs a{};
    a.logpost = s->logpost_blk; a.N = N; a.n = nb; a.i0 = s->blk_i0; a.stabilise = s->opt.theta_stabilise;
    a.seed = s->opt.seed; a.iter = (uint32_t)(s->iter + 1);
    a.theta_out = s->theta_stage; a.degenerate = s->flags + 1; a.err = s->flags;
    return launch_theta_sample(st, a);
}

int launch_beta_on(gpirt_sampler_s* s, hipStream_t st)
{
    BetaArgs a{};
    a.beta = s->beta; a.theta = s->theta; a.y = s->y; a.f = s->f; a.pm = s->pm; a.ps = s->ps;
    a.step = s->step; a.n = s->n; a.m = s->m; a.N = s->N; a.mu = s->mu; a.mu_star = s->mu_star;
    a.seed = s->opt.seed; a.iter = (uint32_t)(s->iter + 1); a.item0 = (uint32_t)s->opt.item0;
    a.err = s->flags;
    if (stream_mode(s)) { a.U = s->U; a.pos = s->pos; a.item_off = s->beta_off; a.cap = s->U_cap; }
    GP_TRY(launch_draw_beta(st, a));
    if (stream_mode(s)) GP_TRY(launch_advance_pos(st, s->pos, s->beta_total));
    return 0;
}

// a deferred draw_beta runs NOW on the main stream; one already running on the sampler's stream is joined
int beta_sync(gpirt_sampler_s* s)
{
    if (s->beta_deferred) {
        s->beta_deferred = false;
        GP_TRY(launch_beta_on(s, s->h->stream));
    }
    if (s->beta_pending) {
        GP_HIP(hipStreamWaitEvent(s->h->stream, s->ev_beta, 0));
        s->beta_pending = false;
    }
    return 0;
}

int do_draw_beta(gpirt_sampler_s* s)
{
    const bool defer = s->h->cfg.early_inv != 2 && !stream_mode(s) && s->haux && s->ev_beta && s->initialised;
    GP_TRY(beta_sync(s));
    if (defer) { s->beta_deferred = true; return 0; }     // do_factor launches it behind the factorisation's last outer panel
    return launch_beta_on(s, s->h->stream);
}

// everything derived from the factor (C, G, cached block inverses on either handle) is stale once L changes
void invalidate_factor_products(gpirt_sampler_s* s)
{
    s->prep_valid = false;
    if (s->haux && s->haux->inv_partial_L == s->L) { s->haux->inv_partial_L = nullptr; s->haux->inv_partial_pairs = 0; }
    if (s->h->trsm_winv_L == s->L) s->h->trsm_winv_L = nullptr;
    if (s->haux && s->haux->trsm_winv_L == s->L) s->haux->trsm_winv_L = nullptr;
}

// the main stream waits for whatever the sampler's own stream still has in flight.  z_too: also a prefill of the next
// draw_f's normals, which is then dropped (every caller but do_factor, whose own prefill it would be)
int aux_join(gpirt_sampler_s* s, bool beta_too = true, bool z_too = true)
{
    hipStream_t st = s->h->stream;
    if (s->prep_pending) { GP_HIP(hipStreamWaitEvent(st, s->ev_prep, 0)); s->prep_pending = false; }
    if (beta_too) GP_TRY(beta_sync(s));
    if (z_too) GP_TRY(z_sync(s));
    return 0;
}

// S = K(theta, theta) + jitter into the lower blocks of L; with the bordered layout also K(c, theta) into the rows below
int build_cov(gpirt_sampler_s* s)
{
    hipStream_t st = s->h->stream;
    GP_TRY(launch_se_kernel_lower(st, s->theta, s->n, s->L, s->ldl, GPIRT_JITTER, s->opt.kernel_fp32 != 0));
    if (s->ext > 0 && !s->ext_grid) GP_TRY(launch_se_kernel(st, s->knodes, s->ext, s->theta, s->n, s->L + s->n, s->ldl, 0.0));
    if (s->ext_grid) GP_TRY(launch_se_kernel(st, s->tstar, s->N, s->theta, s->n, s->L + s->n, s->ldl, 0.0));
    return 0;
}

// the rows below L for the CURRENT (theta, L) by the explicit forward solve (after L or theta was set from outside)
int rebuild_rows(gpirt_sampler_s* s)
{
    if (s->ext == 0 || s->rows_valid) return 0;
    hipStream_t st = s->h->stream;
    const int64_t n = s->n, cols = s->ext_grid ? s->N : s->kr;
    double* Bu = s->rhs;
    GP_TRY(launch_se_kernel(st, s->theta, n, s->ext_grid ? s->tstar : s->knodes, cols, Bu, n, 0.0));
    GP_TRY(launch_trsm_lower(s->h, st, s->L, n, s->ldl, Bu, cols, n, false));
    GP_TRY(launch_transpose(st, Bu, n, cols, n, s->L + n, s->ldl));
    s->rows_valid = true;
    return 0;
}

int do_factor(gpirt_sampler_s* s)
{
    hipStream_t st = s->h->stream;
    GP_TRY(aux_join(s, false, true));     // nothing of the old factor may still be read when it is overwritten (a held-back
                                          // draw_beta does not read it: it is launched below, behind the factorisation)
    invalidate_factor_products(s);
    GP_TRY(build_cov(s));                                                                                  // :76-77
    s->rows_valid = true;
    GP_TRY(launch_potrf_lower(s->h, st, s->L, s->n, s->ldl, false, !s->sticky_info, s->ext)); // :78
    // The low-rank draw_fstar's transposed solve (fstar_prep) goes through the inverses of L's 1024 x 1024 diagonal blocks:
    // ~5 GFLOP to build at n = 8192, and they only need the DIAGONAL blocks, final panel by panel.  Those of every outer
    // panel but the last are built on the sampler's own stream while the last outer panel is factored (16 + 8 whole-CU
    // work-groups and small updates: most of the chip is idle then); only the last block and the solve are left behind.
    const bool early_inv = s->h->cfg.early_inv != 2;
    if (early_inv && s->haux && s->kr > 0 && s->ext > 0 && !s->ext_grid && !stream_mode(s) && s->h->prelast_cols >= 2048) {
        const int64_t p1 = (s->h->prelast_cols / 512) & ~(int64_t)1;
        hipStream_t ax = s->haux->stream;
        GP_HIP(hipStreamWaitEvent(ax, s->h->ev_prelast, 0));
        GP_TRY(trsm_inverses_reserve(s->haux, ax, s->n, s->kr, true));
        s->haux->trsm_winv_L = nullptr;                      // whatever the side handle held is being overwritten
        GP_TRY(trsm_inverses_build(s->haux, ax, s->L, s->n, s->ldl, true, 0, p1));
        s->haux->inv_partial_L = s->L; s->haux->inv_partial_pairs = p1;
    }
    if (early_inv && s->initialised && !s->in_init && s->haux && s->ev_zfill && !stream_mode(s) && s->h->prelast_cols >= 2048) {
        // (not from gpirt_sampler_init, first or repeated: it fills Z itself for the initial f right behind its factorisation)
        // ... and the next draw_f's N(0,1) draws (the factorisation closes iteration s->iter + 1; the next draw_f is
        // iteration s->iter + 2).  Z was last read by this iteration's nu = L z, long finished on the main stream.
        hipStream_t ax = s->haux->stream;
        const uint32_t next_iter = (uint32_t)(s->iter + 2);
        GP_HIP(hipStreamWaitEvent(ax, s->h->ev_prelast, 0));
        GP_TRY(launch_item_uniforms(ax, s->opt.seed, next_iter, GPIRT_ST_F_Z, (uint32_t)s->opt.item0, s->m, s->n, s->Z, true));
        GP_HIP(hipEventRecord(s->ev_zfill, ax));
        s->z_filled_iter = next_iter;
    }
    if (s->beta_deferred) {
        // draw_beta of THIS iteration (do_draw_beta held it back): beside the last outer panel on the sampler's stream
        // when there is such a phase, otherwise right here on the main stream
        s->beta_deferred = false;
        if (s->haux && s->h->prelast_cols >= 2048) {
            hipStream_t ax = s->haux->stream;
            GP_HIP(hipStreamWaitEvent(ax, s->h->ev_prelast, 0));       // (causally behind theta and f on the main stream)
            GP_TRY(launch_beta_on(s, ax));
            GP_HIP(hipEventRecord(s->ev_beta, ax));
            s->beta_pending = true;
        } else {
            GP_TRY(launch_beta_on(s, st));
        }
    }
    s->factor_fresh = true;               // nothing has read this L yet: a hang-guard expiry in it can still be repaired in place
    return 0;
}

// The factorisation enqueued last ended in a hang-guard expiry (info[1]) and nothing has consumed its L yet (factor_fresh):
// theta is intact, so K is rebuilt from it and factored once more with the launch-per-step panel -- the schedule differs,
// the products do not, and the chain goes on as if nothing had happened (L within rounding of the undisturbed factor).
// Whatever was launched behind the bad factor on the sampler's own stream and depends on it (early block inverses) is
// dropped; the Z prefill and a held-back draw_beta do not depend on L and stay.
int recover_factor(gpirt_sampler_s* s)
{
    gpirt_handle_t h = s->h;
    hipStream_t st = h->stream;
    if (s->haux) GP_HIP(hipStreamSynchronize(s->haux->stream));
    GP_TRY(potrf_guard_reset(h, st));
    s->prep_pending = false;
    invalidate_factor_products(s);
    const int saved = h->cfg.panel;
    h->cfg.panel = 2;
    int rc = build_cov(s);
    if (!rc) rc = launch_potrf_lower(h, st, s->L, s->n, s->ldl, false, true, s->ext);
    h->cfg.panel = saved;
    GP_TRY(rc);
    s->rows_valid = true;
    s->factor_fresh = true;
    return 0;
}

// drains the stream and repairs a hang-guard expiry of the factorisation enqueued last (callers that drain it anyway)
int factor_guard_sync(gpirt_sampler_s* s)
{
    gpirt_handle_t h = s->h;
    GP_HIP(hipMemcpyAsync(h->h_info, h->d_info, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    GP_HIP(hipStreamSynchronize(h->stream));
    if (h->h_info[1] != 0 && s->factor_fresh && h->cfg.panel != 2) GP_TRY(recover_factor(s));
    return 0;
}

// draw_theta's CDF came out 0/0 for some respondents: the reference reads theta_star[N] out of bounds there
// (src/draw-theta.cpp:28, quirk Q5); the device wrote NaN and counted them -- the chain cannot go on.
int report_degenerate_theta(gpirt_sampler_s* s, int count)
{
    hipMemsetAsync(s->flags + 1, 0, sizeof(int), s->h->stream);
    set_error("draw_theta: exp() underflowed for %d respondent(s) (0/0 in the reference's CDF, src/draw-theta.cpp:23-34); "
              "theta_stabilise = 1 draws from the same distribution without the underflow", count);
    return GPIRT_E_NUMERIC;
}

inline void mark(gpirt_sampler_s* s, int idx)
{
    if (s->timing) hipEventRecord(s->ev[idx], s->h->stream);
}

}  // namespace

extern "C" {

int gpirt_sampler_create(gpirt_sampler_t* out, gpirt_handle_t h, const double* h_y, int64_t n,
                         int64_t m, const double* h_theta0, const double* h_pm, const double* h_ps,
                         const double* h_step, const gpirt_options* opts, gpirt_rstream_t rs)
{
    GP_ARG(out && h && h_y && h_theta0 && h_pm && h_ps && h_step && n > 0 && m > 0);
    *out = nullptr;
    gpirt_sampler_s* s = new (std::nothrow) gpirt_sampler_s();
    if (!s) { set_error("out of host memory"); return GPIRT_E_ALLOC; }
    s->h = h; s->n = n; s->m = m; s->N = GPIRT_NGRID;
    if (opts) s->opt = *opts; else gpirt_default_options(&s->opt);
    if (s->opt.m_total <= 0) s->opt.m_total = m;
    if (stream_mode(s)) {
        if (!rs) { set_error("GPIRT_RNG_RSTREAM needs an R stream state"); delete s; return GPIRT_E_ARG; }
        if (s->opt.item0 != 0 || s->opt.m_total != m) {
            set_error("R-stream replay is sequential over items and cannot be sharded"); delete s; return GPIRT_E_ARG;
        }
        rstream_sync(rs);            // (another sampler may be running ahead on this stream: it hands it back first)
        s->rs = &rs->r; s->rs_obj = rs;
        rs->attached.push_back(s); rs->forget = rstream_forget;
    }
    const int64_t N = s->N;
    hipStream_t st = h->stream;
    int rc = 0;
#define GP_A(p, cnt) do { rc = dalloc(s, &(p), (size_t)(cnt)); if (rc) { gpirt_sampler_destroy(s); return rc; } } while (0)
    GP_A(s->y, n * m);       GP_A(s->Ypm, n * 2 * m);  GP_A(s->theta, n);       GP_A(s->theta_new, n);
    s->tfd = tf_dims(n, m, N);
    GP_A(s->tf_y8, tf_y8_bytes(s->tfd) / 8 + 2); GP_A(s->tf_gq, tf_gq_bytes(s->tfd) / 8 + 2); GP_A(s->tf_aux, tf_aux_bytes(s->tfd) / 8 + 2);
    GP_A(s->f, n * m);       GP_A(s->Z, n * m);        GP_A(s->NU, n * (m > 1 + TRMV_SPLIT ? m : 1 + TRMV_SPLIT));   GP_A(s->beta, 2 * m);   // (NU: >= 1 + TRMV_SPLIT columns, launch_trmv_lower's parts)
    s->kr = s->opt.kstar_rank;
    if (s->kr != 0 && (!s->opt.fstar_fused || s->kr < 16 || s->kr > 128 || (s->kr % 16) != 0)) {
        set_error("kstar_rank must be a multiple of 16 in 16..128 and needs fstar_fused");
        gpirt_sampler_destroy(s);
        return GPIRT_E_ARG;
    }
    const bool bordered_ok = (n % 64) == 0 && h->cfg.bordered != 2;
    s->ext = (s->kr > 0 && bordered_ok) ? s->kr : 0;
    if (s->kr == 0 && bordered_ok && n >= 1024) {
        s->ext = (N + 63) / 64 * 64;          // 1001 grid rows + 23 rows that stay zero
        s->ext_grid = true;
    }
    s->ldl = n + s->ext;
    GP_A(s->mu, n * m);      GP_A(s->mu_star, N * m + 1); GP_A(s->fstar, N * m + 1); GP_A(s->L, s->ldl * n);
    GP_A(s->tstar, N + 1);   GP_A(s->rhs, n * (N + m) + 2); GP_A(s->mean, N * m + 1); GP_A(s->s, N + 1);
    GP_A(s->Gpm, ((N + 127) / 128 * 128) * 2 * m + 2); GP_A(s->logpost, N * n + 2); GP_A(s->irf_sum, N * m + 1);
    // padding rows stay zero; cleared on the handle's stream (drained below) -- a null-stream hipMemset is not
    // ordered against a non-blocking stream
    if (hipMemsetAsync(s->Gpm, 0, sizeof(double) * (size_t)(((N + 127) / 128 * 128) * 2 * m + 2), st) != hipSuccess) {
        set_error("hipMemsetAsync(Gpm) failed"); gpirt_sampler_destroy(s); return GPIRT_E_HIP;
    }
    GP_A(s->pm, 2 * m);      GP_A(s->ps, 2 * m);       GP_A(s->step, 2 * m);
    GP_A(s->ess_k, m);       GP_A(s->flags, 4);
    if (!s->opt.fstar_fused) GP_A(s->kstar, n * N + 2);
    if (s->kr != 0) {
        GP_A(s->knodes, s->kr); GP_A(s->kV, N * s->kr);
        GP_A(s->kparts, (size_t)KSPLIT * s->kr * (s->kr + m)); GP_A(s->kP, (size_t)s->kr * (s->kr + m));
    }
    if (stream_mode(s)) {
        // consumption of draw_beta per item: rnorm (2 unless step == 0) + runif (1), twice
        std::vector<uint64_t> off((size_t)m);
        uint64_t acc = 0;
        for (int64_t j = 0; j < m; ++j) {
            off[(size_t)j] = acc;
            for (int k = 0; k < 2; ++k) {
                const double sd = h_step[k + 2 * j];
                acc += ((sd > 0.0 && std::isfinite(sd)) ? 2 : 0) + 1;
            }
        }
        s->beta_total = acc;
        s->U_cap = stream_window(s);
        const uint64_t init_need = (uint64_t)m * 2 * (uint64_t)n + 4 * (uint64_t)m + 2 * (uint64_t)N * m + 64;
        if (init_need > s->U_cap) s->U_cap = init_need;
        GP_A(s->U, s->U_cap);       GP_A(s->U2, s->U_cap);      GP_A(s->S, s->U_cap);      GP_A(s->Sraw, s->U_cap);
        GP_A(s->pos, 2);
        // draw_f, three items per pass (rng_ess.hip)
        // The one-phase slice kernel MEETS: its work-groups (320 threads, ~30 KB of LDS) poll each other's flags, so all of
        // them must be resident at once -- at most one per compute unit is what the launch can count on beside foreign work.
        // (The predictor's decide kernel has no such need: a ticket, nobody waits for anybody.)  On a device with fewer
        // compute units than the grid the replay falls back to the item-by-item form.
        s->spec_ok = n >= RS_SPEC_MIN_N && n <= RS3_MAX_N && rs3_slice_wgs(n) <= h->n_cu;
        if (s->spec_ok) {
            const size_t parts = (size_t)((n + RS_KC - 1) / RS_KC);
            const size_t nrm = (size_t)s->U_cap + 8 * (size_t)n + 4096;       // (the predictor's products read up to 8n + 6 + 32 + 16 PD_MAXROUND + 40 past an anchor)
            GP_A(s->Lt, rs_tile_doubles(n));
            GP_A(s->posv, m + 1);    GP_A(s->anchor, 4);    GP_A(s->rs_trace, 128);
            hipMemsetAsync(s->rs_trace, 0, 128 * sizeof(long long), st);
            GP_A(s->rs_flags, 2 * RS3_MAX_WGS);    GP_A(s->rs_partial, 2 * RS3_MAX_WGS * (RS3_TRIALS + 1));
            hipMemsetAsync(s->rs_flags, 0, sizeof(unsigned long long) * 2 * RS3_MAX_WGS, st);
            GP_A(s->Nrm, nrm);       GP_A(s->rs_part, parts * RS3_CAND * (size_t)n);
            hipMemsetAsync(s->Nrm, 0, sizeof(double) * nrm, st);              // (positions no draw has filled are read, never used)
            hipMemsetAsync(s->rs_part, 0, sizeof(double) * parts * RS3_CAND * (size_t)n, st);
            hipMemsetAsync(s->anchor, 0, 4 * sizeof(uint64_t), st);
            const size_t partsP = (size_t)((n + RS3P_KC - 1) / RS3P_KC);
            GP_A(s->Lt32, rs32_tile_floats(n));   GP_A(s->rs_part32, partsP * RS3_CAND * (size_t)n);
            GP_A(s->anchorP, 8);     GP_A(s->rs_ctl, 8);     GP_A(s->rs_posP, 2);
            GP_A(s->rs_dec_part, (size_t)RS3_CAND * 8 * 17 + 8);  GP_A(s->rs_dec_rec, (size_t)RS3_CAND * 20 + 8);  GP_A(s->rs_dec_ticket, 32 * 9);      // the top word + one per row part, 128 bytes apart
            hipMemsetAsync(s->rs_dec_ticket, 0, 32 * 9 * sizeof(unsigned), st);
            GP_A(s->rs_kpred, m);    GP_A(s->rs_kv, m);      GP_A(s->rs_used, m);     GP_A(s->rs_ierr, m);    GP_A(s->rs_errP, 4);
            hipMemsetAsync(s->rs_part32, 0, sizeof(float) * partsP * RS3_CAND * (size_t)n, st);
            hipMemsetAsync(s->anchorP, 0, 8 * sizeof(uint64_t), st);
            hipMemsetAsync(s->rs_ctl, 0, 8 * sizeof(uint64_t), st);
            hipMemsetAsync(s->rs_errP, 0, 4 * sizeof(int), st);
            std::vector<uint32_t> units;
            rs3_unit_table(n, units, &s->rs_nfull);
            s->rs_nunits = (int)units.size();
            GP_A(s->rs_units, units.size());
            hipMemcpyAsync(s->rs_units, units.data(), units.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
            std::vector<uint32_t> unitsP;
            rs3p_unit_table(n, unitsP, &s->rs_nfullP);
            s->rs_nunitsP = (int)unitsP.size();
            GP_A(s->rs_unitsP, unitsP.size());
            hipMemcpyAsync(s->rs_unitsP, unitsP.data(), unitsP.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
            // the structured form of the predictor's pass (rs_lr.hip): from 4096 rows on (below, a pass over L is a few MB anyway: 2048 x 256 runs at 270 it/s with it, 298 without),
            // not with the single-precision kernel build (S then is not the kernel the basis represents, to 6e-8)
            s->lr_on = h->cfg.rs_lr != 2 && n >= 4096 && !s->opt.kernel_fp32;
            std::vector<uint32_t> unitsL;
            if (s->lr_on) {
                std::vector<double> nodes, wts, Mn;
                rs_lr_nodes(nodes, wts, Mn);
                const size_t nb = (size_t)((n + 63) / 64), lparts = (size_t)((n + RS3P_KC - 1) / RS3P_KC);
                GP_A(s->lr_nodes, RS_LR_RANK);  GP_A(s->lr_wts, RS_LR_RANK);  GP_A(s->lr_M, RS_LR_RANK * RS_LR_RANK);
                GP_A(s->lr_V64, nb * 64 * RS_LR_RANK);  GP_A(s->lr_Gb, nb * RS_LR_RANK * RS_LR_RANK);
                GP_A(s->lr_V32t, (size_t)RS_LR_RANK * (size_t)n);  GP_A(s->lr_Ct32, (size_t)(RS_LR_RANK / RS_ROWS) * (size_t)rs32_tile_octs(n) * 256);
                GP_A(s->lr_Y, lparts * RS3_CAND * RS_LR_RANK);  GP_A(s->lr_bad, 4);
                hipMemcpyAsync(s->lr_nodes, nodes.data(), sizeof(double) * RS_LR_RANK, hipMemcpyHostToDevice, st);
                hipMemcpyAsync(s->lr_wts, wts.data(), sizeof(double) * RS_LR_RANK, hipMemcpyHostToDevice, st);
                hipMemcpyAsync(s->lr_M, Mn.data(), sizeof(double) * RS_LR_RANK * RS_LR_RANK, hipMemcpyHostToDevice, st);
                hipMemsetAsync(s->lr_Ct32, 0, sizeof(float) * (size_t)(RS_LR_RANK / RS_ROWS) * (size_t)rs32_tile_octs(n) * 256, st);
                hipMemsetAsync(s->lr_Y, 0, sizeof(float) * lparts * RS3_CAND * RS_LR_RANK, st);
                hipMemsetAsync(s->lr_bad, 0, 4 * sizeof(int), st);
                rs_lr_unit_table(n, unitsL);
                s->lr_nunits = (int)unitsL.size();
                GP_A(s->lr_units, unitsL.size());
                hipMemcpyAsync(s->lr_units, unitsL.data(), unitsL.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
            }
            hipStreamSynchronize(st);                                          // (unitsP, unitsL leave scope with units below)
            hipStreamSynchronize(st);                                          // (units leaves scope)
        }
        GP_A(s->beta_off, m);
        GP_A(s->fstar_off, N + 8);
        if (hipHostMalloc(&s->hU, s->U_cap * sizeof(double), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&s->hA, s->U_cap * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&s->h_pos, 2 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess ||
            hipStreamCreateWithFlags(&s->cs, hipStreamNonBlocking) != hipSuccess ||
            hipHostMalloc(&s->h_next, 8 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_up, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_asm, hipEventDisableTiming) != hipSuccess) {
            set_error("pinned allocation for the R-stream window failed");
            gpirt_sampler_destroy(s);
            return GPIRT_E_ALLOC;
        }
        hipMemcpyAsync(s->beta_off, off.data(), sizeof(uint64_t) * (size_t)m, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
    }
#undef GP_A
    if (hipHostMalloc(&s->h_flags, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        set_error("pinned allocation failed"); gpirt_sampler_destroy(s); return GPIRT_E_ALLOC;
    }
    for (int i = 0; i <= ST_COUNT; ++i) hipEventCreate(&s->ev[i]);
    // uploads (host buffers are caller-owned and never modified: R semantics)
    std::vector<double> ts((size_t)N);
    for (int64_t i = 0; i < N; ++i) ts[(size_t)i] = -5.0 + (double)i * 0.01;   // src/gpirtMCMC.cpp:35
    hipMemcpyAsync(s->y, h_y, sizeof(double) * (size_t)(n * m), hipMemcpyHostToDevice, st);
    hipMemcpyAsync(s->theta, h_theta0, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(s->pm, h_pm, sizeof(double) * (size_t)(2 * m), hipMemcpyHostToDevice, st);
    hipMemcpyAsync(s->ps, h_ps, sizeof(double) * (size_t)(2 * m), hipMemcpyHostToDevice, st);
    hipMemcpyAsync(s->step, h_step, sizeof(double) * (size_t)(2 * m), hipMemcpyHostToDevice, st);
    hipMemcpyAsync(s->tstar, ts.data(), sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st);
    std::vector<double> nodes, V;
    if (s->kr > 0) {
        // Chebyshev interpolation of  t -> exp(-(theta - t)^2 / 2)  on the grid's interval [-5, 5]:
        //   K(theta_i, t_j) = sum_k K(theta_i, c_k) V[j][k],  V = barycentric Lagrange basis at the r first-kind
        // Chebyshev points c_k, evaluated at the 1001 grid points in long double.  The integrand is entire and
        // the interval is 10 length-scales wide: r = 56 already reaches 1.3e-15 max-abs error (Lebesgue
        // constant 3.6), so K*^T (n x 1001) is replaced by its exact rank-r factorisation U V^T, U = K(theta, c).
        const int r = s->kr;
        const long double pi = 3.141592653589793238462643383279502884L;
        std::vector<long double> c((size_t)r), w((size_t)r);
        nodes.resize((size_t)r); V.assign((size_t)(N * r), 0.0);
        for (int k = 0; k < r; ++k) {
            const long double a = (2 * k + 1) * pi / (2 * r);
            c[(size_t)k] = 5.0L * cosl(a);
            w[(size_t)k] = ((k & 1) ? -1.0L : 1.0L) * sinl(a);
            nodes[(size_t)k] = (double)c[(size_t)k];
        }
        for (int64_t j = 0; j < N; ++j) {
            const long double t = (long double)ts[(size_t)j];
            int hit = -1;
            long double den = 0.0L;
            for (int k = 0; k < r; ++k) {
                const long double d = t - (long double)nodes[(size_t)k];     // the nodes as the device will see them
                if (d == 0.0L) { hit = k; break; }
                den += w[(size_t)k] / d;
            }
            for (int k = 0; k < r; ++k) {
                long double v;
                if (hit >= 0) v = (k == hit) ? 1.0L : 0.0L;
                else v = (w[(size_t)k] / (t - (long double)nodes[(size_t)k])) / den;
                V[(size_t)(j + k * N)] = (double)v;
            }
        }
        hipMemcpyAsync(s->knodes, nodes.data(), sizeof(double) * (size_t)r, hipMemcpyHostToDevice, st);
        hipMemcpyAsync(s->kV, V.data(), sizeof(double) * (size_t)(N * r), hipMemcpyHostToDevice, st);
    }
    hipMemsetAsync(s->L, 0, sizeof(double) * (size_t)(s->ldl * n), st);    // strict upper stays zero
    hipMemsetAsync(s->irf_sum, 0, sizeof(double) * (size_t)(N * m), st);   // :42
    hipMemsetAsync(s->flags, 0, 4 * sizeof(int), st);
    hipMemsetAsync(s->ess_k, 0, sizeof(int) * (size_t)m, st);
    launch_indicators(st, s->y, n, m, s->Ypm);
    launch_tf_indicators(st, s->y, n, n, m, s->tfd, s->tf_y8);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        set_error("sampler upload failed"); gpirt_sampler_destroy(s); return GPIRT_E_HIP;
    }
    if (!stream_mode(s)) {
        if (!h->aux) {
            if (create_side_handle(&h->aux, h->device) != 0) { gpirt_sampler_destroy(s); return GPIRT_E_HIP; }
            h->aux->cfg = h->cfg;
        }
        s->haux = h->aux;
        if (
            hipEventCreateWithFlags(&s->ev_trmm, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_prep, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_zfill, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_beta, hipEventDisableTiming) != hipSuccess) {
            gpirt_sampler_destroy(s);
            return GPIRT_E_HIP;
        }
    }
    // keep pm/ps on the host for the R-stream init of beta
    s->host_tmp.assign(h_pm, h_pm + 2 * m);
    s->host_tmp.insert(s->host_tmp.end(), h_ps, h_ps + 2 * m);
    h->live_samplers += 1;
    s->counted = true;
    *out = s;
    return 0;
}

int gpirt_sampler_destroy(gpirt_sampler_t s)
{
    if (!s) return 0;
    if (s->h) hipStreamSynchronize(s->h->stream);
    if (s->rs_obj && s->rs_obj->owner == s) ahead_resolve(s, false);     // the caller's generator goes back to the consumed position
    if (s->rs_obj) {
        auto& at = s->rs_obj->attached;
        for (size_t q = 0; q < at.size(); ++q) if (at[q] == s) { at.erase(at.begin() + (std::ptrdiff_t)q); break; }
    }
    if (s->cs) { hipStreamSynchronize(s->cs); hipStreamDestroy(s->cs); }
    if (s->h_next) hipHostFree(s->h_next);
    if (s->ev_up) hipEventDestroy(s->ev_up);
    if (s->ev_asm) hipEventDestroy(s->ev_asm);
    if (s->hA) hipHostFree(s->hA);
    if (s->haux) {
        hipStreamSynchronize(s->haux->stream);       // (the side handle belongs to the main handle: not destroyed here)
        if (s->haux->trsm_winv_L == s->L) s->haux->trsm_winv_L = nullptr;
        if (s->haux->inv_partial_L == s->L) { s->haux->inv_partial_L = nullptr; s->haux->inv_partial_pairs = 0; }
    }
    if (s->h && s->h->trsm_winv_L == s->L) s->h->trsm_winv_L = nullptr;
    if (s->ev_trmm) hipEventDestroy(s->ev_trmm);
    if (s->ev_prep) hipEventDestroy(s->ev_prep);
    if (s->ev_zfill) hipEventDestroy(s->ev_zfill);
    if (s->ev_beta) hipEventDestroy(s->ev_beta);
    for (void* p : s->allocs) hipFree(p);
    if (s->hU) hipHostFree(s->hU);
    if (s->h_pos) hipHostFree(s->h_pos);
    if (s->h_flags) hipHostFree(s->h_flags);
    for (int i = 0; i <= ST_COUNT; ++i) if (s->ev[i]) hipEventDestroy(s->ev[i]);
    gpirt_handle_t h = s->h;
    const bool counted = s->counted;
    delete s;
    if (h && counted) {
        h->live_samplers -= 1;
        if (h->zombie && h->live_samplers == 0) { h->zombie = false; gpirt_destroy(h); }    // the handle outlived by its samplers
    }
    return 0;
}

// src/gpirtMCMC.cpp:13-47
int gpirt_sampler_init(gpirt_sampler_t s)
{
    GP_ARG(s != nullptr);
    gpirt_handle_t h = s->h;
    hipStream_t st = h->stream;
    const int64_t n = s->n, m = s->m, N = s->N;
    GP_TRY(aux_join(s));                  // (a repeated init: nothing of the previous chain may still be in flight)
    if (stream_mode(s)) {
        if (!s->rs) { set_error("the R stream of this sampler has been destroyed"); return GPIRT_E_ARG; }
        rstream_sync(s->rs_obj);          // init draws from the generator directly: from the consumed position
    }
    s->in_init = true;
    const int rc_f = do_factor(s);                                                // :15-17
    s->in_init = false;
    GP_TRY(rc_f);
    GP_TRY(factor_guard_sync(s));         // (init drains the stream below anyway)
    if (!stream_mode(s)) {
        GP_TRY(launch_item_uniforms(st, s->opt.seed, 0, GPIRT_ST_INIT_F, (uint32_t)s->opt.item0, m, n, s->Z, true));
        GP_TRY(launch_gemm(h, st, false, false, TRI_A_LOWER, n, m, n, 1.0, s->L, s->ldl, s->Z, n, 0.0, s->f, n)); // :18-21
        // beta(p, j) = R::rnorm(prior_mean, prior_sd), index = p                    :22-27
        std::vector<double> b((size_t)(2 * m));
        const double* pm = s->host_tmp.data();
        const double* ps = pm + 2 * m;
        for (int64_t j = 0; j < m; ++j)
            for (int p = 0; p < 2; ++p) {
                const double mu = pm[p + 2 * j], sd = ps[p + 2 * j];
                double v;
                if (mu != mu || !std::isfinite(sd) || sd < 0.0) v = NAN;
                else if (sd == 0.0 || !std::isfinite(mu)) v = mu;
                else v = mu + sd * qnorm_as241(item_uniform(s->opt.seed, 0, GPIRT_ST_INIT_BETA,
                                                            (uint32_t)(s->opt.item0 + j), (uint32_t)p));
                b[(size_t)(p + 2 * j)] = v;
            }
        GP_HIP(hipMemcpyAsync(s->beta, b.data(), sizeof(double) * b.size(), hipMemcpyHostToDevice, st));
        GP_HIP(hipStreamSynchronize(st));
    } else {
        // window layout: [f init: m x 2n][beta init: <= 4m][fstar: 2 N m]
        const uint64_t nf = (uint64_t)m * 2 * (uint64_t)n;
        // beta consumes on the host, directly behind the f-init block
        s->saved = *s->rs;
        s->rs->fill_unif(s->hU, nf);
        std::vector<double> b((size_t)(2 * m));
        const double* pm = s->host_tmp.data();
        const double* ps = pm + 2 * m;
        uint64_t nb = 0;
        for (int64_t j = 0; j < m; ++j)
            for (int p = 0; p < 2; ++p) {
                const double mu = pm[p + 2 * j], sd = ps[p + 2 * j];
                double v;
                if (mu != mu || !std::isfinite(sd) || sd < 0.0) v = NAN;
                else if (sd == 0.0 || !std::isfinite(mu)) v = mu;
                else { v = mu + sd * s->rs->norm(); nb += 2; }
                b[(size_t)(p + 2 * j)] = v;
            }
        const uint64_t nfs = 2 * (uint64_t)N * (uint64_t)m;
        s->rs->fill_unif(s->hU + nf, nfs);
        (void)nb;
        GP_HIP(hipMemcpyAsync(s->U, s->hU, (nf + nfs) * sizeof(double), hipMemcpyHostToDevice, st));
        GP_HIP(hipMemsetAsync(s->pos, 0, sizeof(uint64_t), st));
        GP_TRY(launch_rstream_normals(st, s->U, s->pos, 2 * n, n, m, s->Z));
        GP_TRY(launch_gemm(h, st, false, false, TRI_A_LOWER, n, m, n, 1.0, s->L, s->ldl, s->Z, n, 0.0, s->f, n));
        GP_TRY(launch_advance_pos(st, s->pos, nf));
        GP_HIP(hipMemcpyAsync(s->beta, b.data(), sizeof(double) * b.size(), hipMemcpyHostToDevice, st));
        GP_HIP(hipStreamSynchronize(st));
    }
    GP_TRY(launch_linear_mean(st, s->theta, n, s->beta, m, s->mu));               // :30-33
    GP_TRY(launch_linear_mean(st, s->tstar, N, s->beta, m, s->mu_star));          // :37-40
    GP_TRY(do_draw_fstar(s, 0));                                                  // :41
    if (stream_mode(s)) {
        // the fstar block may have consumed fewer than 2 N m uniforms (s_i <= 0): rewind exactly
        GP_HIP(hipMemcpyAsync(s->h_pos, s->pos, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        GP_HIP(hipStreamSynchronize(st));
        const uint64_t nf = (uint64_t)m * 2 * (uint64_t)n;
        const uint64_t used_fstar = *s->h_pos - nf;
        // host state currently sits after [f][beta][2Nm]; recompute: after [f][beta] + used_fstar
        RStream r = s->saved;
        r.skip(nf);
        const double* pm = s->host_tmp.data();
        const double* ps = pm + 2 * m;
        for (int64_t j = 0; j < m; ++j)
            for (int p = 0; p < 2; ++p) {
                const double mu = pm[p + 2 * j], sd = ps[p + 2 * j];
                if (!(mu != mu || !std::isfinite(sd) || sd < 0.0) && !(sd == 0.0 || !std::isfinite(mu))) {
                    (void)r.next32(); (void)r.next32();
                }
            }
        r.skip(used_fstar);
        *s->rs = r;
    }
    s->iter = 0;
    s->initialised = true;
    return 0;
}

int gpirt_sampler_draw_f(gpirt_sampler_t s) { GP_ARG(s && s->initialised); return do_draw_f(s); }
int gpirt_sampler_draw_fstar(gpirt_sampler_t s) { GP_ARG(s && s->initialised); return do_draw_fstar(s, (uint32_t)(s->iter + 1)); }
int gpirt_sampler_theta_partial(gpirt_sampler_t s) { GP_ARG(s && s->initialised); return do_theta_partial(s); }
int gpirt_sampler_theta_finish(gpirt_sampler_t s) { GP_ARG(s && s->initialised); return do_theta_finish(s); }

int gpirt_sampler_set_theta_block(gpirt_sampler_t s, const double* y_block, int64_t i0, int64_t n_block, int64_t m_total)
{
    GP_ARG(s && (y_block || n_block == 0) && i0 >= 0 && n_block >= 0 && i0 + n_block <= s->n && m_total >= s->m);
    if (stream_mode(s)) { set_error("the R-stream replay cannot be sharded"); return GPIRT_E_ARG; }
    if (s->fstar_full) { set_error("the theta block is already set"); return GPIRT_E_ARG; }
    const int64_t N = s->N, Np = (N + 127) / 128 * 128;
    int rc = 0;
    double* yb = nullptr;
#define GP_B(p, cnt) do { rc = dalloc(s, &(p), (size_t)(cnt)); if (rc) return rc; } while (0)
    GP_B(s->Ypm_blk, n_block * 2 * m_total + 2); GP_B(s->Gpm_full, Np * 2 * m_total + 2);
    GP_B(s->logpost_blk, N * n_block + 2);       GP_B(s->fstar_full, N * m_total + 2);
    GP_B(s->theta_stage, s->n + 1);              GP_B(yb, n_block * m_total + 1);
    s->tfd_blk = tf_dims(n_block, m_total, N);
    GP_B(s->tf_y8_blk, tf_y8_bytes(s->tfd_blk) / 8 + 2); GP_B(s->tf_gq_blk, tf_gq_bytes(s->tfd_blk) / 8 + 2);
    GP_B(s->tf_aux_blk, tf_aux_bytes(s->tfd_blk) / 8 + 2);
#undef GP_B
    hipStream_t st = s->h->stream;
    GP_HIP(hipMemsetAsync(s->Gpm_full, 0, sizeof(double) * (size_t)(Np * 2 * m_total + 2), st));   // padding rows stay zero
    if (n_block > 0) {
        GP_HIP(hipMemcpyAsync(yb, y_block, sizeof(double) * (size_t)(n_block * m_total), hipMemcpyHostToDevice, st));
        GP_TRY(launch_indicators(st, yb, n_block, m_total, s->Ypm_blk));
        GP_TRY(launch_tf_indicators(st, yb, n_block, n_block, m_total, s->tfd_blk, s->tf_y8_blk));
    }
    GP_HIP(hipStreamSynchronize(st));
    s->blk_i0 = i0; s->blk_n = n_block; s->blk_m = m_total;
    return 0;
}

int gpirt_sampler_theta_block(gpirt_sampler_t s)
{
    GP_ARG(s && s->initialised);
    if (!s->fstar_full) { set_error("gpirt_sampler_set_theta_block has not been called"); return GPIRT_E_ARG; }
    return do_theta_block(s);
}

/* theta := the staged draw (after the blocks of all ranks have been summed into "theta_stage") */
int gpirt_sampler_theta_commit(gpirt_sampler_t s)
{
    GP_ARG(s && s->initialised && s->theta_stage);
    GP_HIP(hipMemcpyAsync(s->theta, s->theta_stage, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToDevice, s->h->stream));
    return 0;
}
int gpirt_sampler_draw_beta(gpirt_sampler_t s) { GP_ARG(s && s->initialised); return do_draw_beta(s); }
int gpirt_sampler_factor(gpirt_sampler_t s)
{
    GP_ARG(s && s->initialised);
    GP_TRY(do_factor(s));
    s->iter += 1;            // the factorisation closes an iteration (src/gpirtMCMC.cpp:78,97)
    return 0;
}

int gpirt_sampler_build_cov(gpirt_sampler_t s)
{
    GP_ARG(s && s->initialised);
    hipStream_t st = s->h->stream;
    if (!s->sticky_info) GP_HIP(hipMemsetAsync(s->h->d_info, 0, sizeof(int), st));
    GP_TRY(aux_join(s));
    invalidate_factor_products(s);
    return build_cov(s);                                                                                   // :76-77
}

// L arrived from elsewhere (a broadcast, the distributed pieces, gpirt_sampler_set): closes the iteration without
// factoring.  rows_with_L says whether the rows below the n x n factor (bordered layout, gpirt_sampler_ldl) arrived
// with it -- the WHOLE ldl x n buffer was received, as gpirt_amd/distributed.py does.  If not (a host that moved only
// the n x n factor), they still belong to the previous (theta, L) and are rebuilt by the explicit forward solve before
// draw_fstar reads them.
int gpirt_sampler_adopt_factor(gpirt_sampler_t s, int rows_with_L)
{
    GP_ARG(s && s->initialised);
    GP_TRY(beta_sync(s));                // a held-back draw_beta belongs to the iteration that closes here
    invalidate_factor_products(s);
    s->factor_fresh = false;             // (nothing here could rebuild a factor that arrived from elsewhere)
    s->rows_valid = rows_with_L != 0;
    s->iter += 1;
    return 0;
}

int gpirt_sampler_skip_factor(gpirt_sampler_t s) { return gpirt_sampler_adopt_factor(s, 0); }

int gpirt_sampler_step(gpirt_sampler_t s)
{
    GP_ARG(s && s->initialised);
    if (stream_mode(s)) GP_TRY(stream_begin(s, stream_window(s)));
    mark(s, 0);
    GP_TRY(do_draw_f(s));            mark(s, 1);
    GP_TRY(do_draw_fstar(s, (uint32_t)(s->iter + 1))); mark(s, 2);
    GP_TRY(do_theta_partial(s));     mark(s, 3);
    GP_TRY(do_theta_finish(s));      mark(s, 4);
    GP_TRY(do_draw_beta(s));         mark(s, 5);
    GP_TRY(do_factor(s));            mark(s, 6);
    s->iter += 1;
    if (stream_mode(s)) GP_TRY(stream_end(s, stream_window(s)));
    if (s->timing) {
        GP_HIP(hipEventSynchronize(s->ev[ST_COUNT]));
        for (int i = 0; i < ST_COUNT; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]);
            s->stage_ms[i] = ms;
        }
    }
    return 0;
}

int gpirt_sampler_accumulate_irf(gpirt_sampler_t s)
{
    GP_ARG(s && s->initialised);
    return launch_axpy_irf(s->h->stream, s->irf_sum, s->fstar, s->N * s->m);     // :103
}

int gpirt_sampler_iteration(gpirt_sampler_t s, int* iter)
{
    GP_ARG(s && iter);
    *iter = s->iter;
    return 0;
}

int gpirt_sampler_set_iteration(gpirt_sampler_t s, int iter)
{
    GP_ARG(s && s->initialised && iter >= 0);
    if (stream_mode(s)) { set_error("the R-stream replay has no iteration-keyed sub-streams"); return GPIRT_E_ARG; }
    GP_TRY(aux_join(s));
    s->iter = iter;
    return 0;
}

int gpirt_sampler_check(gpirt_sampler_t s)
{
    GP_ARG(s != nullptr);
    gpirt_handle_t h = s->h;
    hipStream_t st = h->stream;
    GP_TRY(aux_join(s));                  // the flags of work still running on the sampler's own stream
    GP_HIP(hipMemcpyAsync(h->h_info, h->d_info, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    GP_HIP(hipMemcpyAsync(s->h_flags, s->flags, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    GP_HIP(hipStreamSynchronize(st));
    if (h->h_info[1] != 0) {
        // hang-guard expiry.  Repairable in place while nothing has read the bad L (the usual case: check() right behind a
        // step) and the launch-per-step panel was not what failed; otherwise the chain has already consumed garbage.
        if (!s->factor_fresh || h->cfg.panel == 2) return report_panel_guard(h, h->h_info, st);
        GP_TRY(recover_factor(s));
        GP_HIP(hipMemcpyAsync(h->h_info, h->d_info, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
        GP_HIP(hipStreamSynchronize(st));
        if (h->h_info[1] != 0) return report_panel_guard(h, h->h_info, st);
    }
    if (*h->h_info > 0) {
        set_error("chol(): decomposition failed (leading minor of order %d is not positive definite)", *h->h_info);
        return *h->h_info;
    }
    if (s->h_flags[0] != 0) {
        set_error("sampler state is not finite (flag %d): elliptical slice sampler did not terminate or the R stream window overflowed", s->h_flags[0]);
        return s->h_flags[0];
    }
    if (s->h_flags[1] != 0) return report_degenerate_theta(s, s->h_flags[1]);
    return 0;
}

static int lookup(gpirt_sampler_t s, const char* name, void** p, int64_t* count)
{
    const int64_t n = s->n, m = s->m, N = s->N;
    struct E { const char* k; void* p; int64_t c; } tab[] = {
        { "theta", s->theta, n }, { "f", s->f, n * m }, { "beta", s->beta, 2 * m }, { "mu", s->mu, n * m },
        { "mu_star", s->mu_star, N * m }, { "fstar", s->fstar, N * m }, { "L", s->L, n * n },
        { "logpost", s->logpost, N * n }, { "irf_sum", s->irf_sum, N * m }, { "ess_k", s->ess_k, m },
        { "s", s->s, N }, { "mean", s->mean, N * m }, { "nu", s->NU, n * m }, { "z", s->Z, n * m },
        { "y", s->y, n * m }, { "rs_trace", s->rs_trace, s->rs_trace ? 128 : 0 },
        { "rs_stats", s->rs_ctl, s->rs_ctl ? 8 : 0 },     // 64-bit words: [first item not committed, mispredictions found, ...]
        { "fstar_full", s->fstar_full, s->fstar_full ? N * s->blk_m : 0 }, { "theta_stage", s->theta_stage, s->theta_stage ? n : 0 },
    };
    for (auto& e : tab)
        if (strcmp(e.k, name) == 0) { *p = e.p; *count = e.c; return 0; }
    set_error("unknown sampler array '%s'", name);
    return GPIRT_E_ARG;
}

int gpirt_sampler_devptr(gpirt_sampler_t s, const char* name, void** d_ptr, int64_t* count)
{
    GP_ARG(s && name && d_ptr && count);
    GP_TRY(lookup(s, name, d_ptr, count));
    if (strcmp(name, "L") == 0) *count = s->ldl * s->n;     // the whole buffer (leading dimension gpirt_sampler_ldl)
    return 0;
}

int gpirt_sampler_ldl(gpirt_sampler_t s, int64_t* ldl)
{
    GP_ARG(s && ldl);
    *ldl = s->ldl;
    return 0;
}

// ---- the factorisation in pieces on this sampler's L (distributed hosts; include/gpirt_hip.h) ---------------------
int gpirt_sampler_panel_factor(gpirt_sampler_t s, int64_t p)
{
    GP_ARG(s && s->initialised);
    return potrf_panel_factor(s->h, s->h->stream, s->L, s->n, s->ldl, p, s->ext);
}

int gpirt_sampler_panel_update(gpirt_sampler_t s, int64_t p, int64_t c)
{
    GP_ARG(s && s->initialised);
    return potrf_panel_update(s->h, s->h->stream, s->L, s->n, s->ldl, p, c, s->ext);
}

int gpirt_sampler_panel_copy(gpirt_sampler_t s, int64_t p, double* d_buf, int to_buf)
{
    GP_ARG(s && s->initialised && d_buf);
    return potrf_panel_copy(s->h->stream, s->L, s->n, s->ldl, p, d_buf, to_buf != 0, s->ext);
}

int gpirt_sampler_panel_rows(gpirt_sampler_t s, int64_t* rows)
{
    GP_ARG(s && rows);
    *rows = s->n + s->ext;
    return 0;
}

// ... and by halves of an outer panel (first sub-panel / the rest), for a host that pipelines the broadcasts
int gpirt_sampler_panel_factor_part(gpirt_sampler_t s, int64_t p, int half)
{
    GP_ARG(s && s->initialised);
    return potrf_panel_factor(s->h, s->h->stream, s->L, s->n, s->ldl, p, s->ext, half);
}

int gpirt_sampler_panel_update_part(gpirt_sampler_t s, int64_t p, int64_t c, int part)
{
    GP_ARG(s && s->initialised);
    return potrf_panel_update(s->h, s->h->stream, s->L, s->n, s->ldl, p, c, s->ext, part);
}

int gpirt_sampler_panel_copy_part(gpirt_sampler_t s, int64_t p, int half, double* d_buf, int64_t buf_doubles, int to_buf)
{
    GP_ARG(s && s->initialised && d_buf && buf_doubles >= 0);
    return potrf_panel_copy(s->h->stream, s->L, s->n, s->ldl, p, d_buf, to_buf != 0, s->ext, half, buf_doubles);
}

// dst's chain state := src's (theta, f, beta, mu, mu_star, fstar, the n x n factor, the iteration counter): lets a second
// sampler with other options replay a stage on the same state (bench.py's in-run check of the draw_fstar forms)
int gpirt_sampler_copy_state(gpirt_sampler_t dst, gpirt_sampler_t src)
{
    GP_ARG(dst && src && dst->initialised && src->initialised && dst->n == src->n && dst->m == src->m && dst->h == src->h);
    hipStream_t st = dst->h->stream;
    const int64_t n = dst->n, m = dst->m, N = dst->N;
    GP_TRY(aux_join(dst)); GP_TRY(aux_join(src));
    src->factor_fresh = false;
    invalidate_factor_products(dst);
    GP_HIP(hipMemcpyAsync(dst->theta, src->theta, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    GP_HIP(hipMemcpyAsync(dst->f, src->f, sizeof(double) * (size_t)(n * m), hipMemcpyDeviceToDevice, st));
    GP_HIP(hipMemcpyAsync(dst->beta, src->beta, sizeof(double) * (size_t)(2 * m), hipMemcpyDeviceToDevice, st));
    GP_HIP(hipMemcpyAsync(dst->mu, src->mu, sizeof(double) * (size_t)(n * m), hipMemcpyDeviceToDevice, st));
    GP_HIP(hipMemcpyAsync(dst->mu_star, src->mu_star, sizeof(double) * (size_t)(N * m), hipMemcpyDeviceToDevice, st));
    GP_HIP(hipMemcpyAsync(dst->fstar, src->fstar, sizeof(double) * (size_t)(N * m), hipMemcpyDeviceToDevice, st));
    GP_HIP(hipMemcpy2DAsync(dst->L, (size_t)dst->ldl * 8, src->L, (size_t)src->ldl * 8, (size_t)n * 8, (size_t)n,
                            hipMemcpyDeviceToDevice, st));
    dst->rows_valid = false;
    GP_TRY(rebuild_rows(dst));            // the rows below L for the copied (theta, L)
    dst->iter = src->iter;
    return 0;
}

int gpirt_sampler_get(gpirt_sampler_t s, const char* name, double* h_out, int64_t count)
{
    GP_ARG(s && name && h_out);
    void* p; int64_t c;
    GP_TRY(lookup(s, name, &p, &c));
    GP_ARG(count <= c);
    const size_t esz = strcmp(name, "ess_k") == 0 ? sizeof(int) : sizeof(double);
    GP_TRY(aux_join(s));
    if (strcmp(name, "L") == 0 && s->ldl != s->n) {       // n x n out of the (n + ext) x n buffer
        GP_ARG(count == s->n * s->n);
        GP_HIP(hipMemcpy2DAsync(h_out, (size_t)s->n * 8, p, (size_t)s->ldl * 8, (size_t)s->n * 8, (size_t)s->n,
                                hipMemcpyDeviceToHost, s->h->stream));
    } else {
        GP_HIP(hipMemcpyAsync(h_out, p, esz * (size_t)count, hipMemcpyDeviceToHost, s->h->stream));
    }
    GP_HIP(hipStreamSynchronize(s->h->stream));
    return 0;
}

int gpirt_sampler_set(gpirt_sampler_t s, const char* name, const double* h_in, int64_t count)
{
    GP_ARG(s && name && h_in);
    void* p; int64_t c;
    GP_TRY(lookup(s, name, &p, &c));
    GP_ARG(count <= c && strcmp(name, "ess_k") != 0);
    GP_TRY(aux_join(s));
    invalidate_factor_products(s);
    if (strcmp(name, "L") == 0 || strcmp(name, "theta") == 0) s->rows_valid = false;
    if (strcmp(name, "L") == 0 && s->ldl != s->n) {
        GP_ARG(count == s->n * s->n);
        GP_HIP(hipMemcpy2DAsync(p, (size_t)s->ldl * 8, h_in, (size_t)s->n * 8, (size_t)s->n * 8, (size_t)s->n,
                                hipMemcpyHostToDevice, s->h->stream));
    } else {
        GP_HIP(hipMemcpyAsync(p, h_in, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s->h->stream));
    }
    GP_HIP(hipStreamSynchronize(s->h->stream));
    return 0;
}

// IRFs *= 1/S ; plogis : src/gpirtMCMC.cpp:106-111 (Q7: S = 0 gives NaN, as in the reference)
int gpirt_sampler_finish_irfs(gpirt_sampler_t s, int sample_iterations, double* h_irfs)
{
    GP_ARG(s && h_irfs);
    const int64_t cnt = s->N * s->m;
    GP_HIP(hipMemcpyAsync(h_irfs, s->irf_sum, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, s->h->stream));
    GP_HIP(hipStreamSynchronize(s->h->stream));
    const double inv = 1.0 / (double)sample_iterations;
    for (int64_t i = 0; i < cnt; ++i) h_irfs[i] = 1.0 / (1.0 + exp(-(h_irfs[i] * inv)));
    return 0;
}

int gpirt_sampler_enable_timing(gpirt_sampler_t s, int on)
{
    GP_ARG(s != nullptr);
    s->timing = on != 0;
    return 0;
}

int gpirt_sampler_stage_times(gpirt_sampler_t s, double* ms_out, int max_stages, int* n_stages,
                              const char** names_out)
{
    GP_ARG(s && ms_out && n_stages);
    static const char names[] = "draw_f\0draw_fstar\0theta_gemm\0theta_sample\0draw_beta\0factor\0";
    const int k = max_stages < ST_COUNT ? max_stages : ST_COUNT;
    for (int i = 0; i < k; ++i) ms_out[i] = s->stage_ms[i];
    *n_stages = k;
    if (names_out) *names_out = names;
    (void)kStageNames;
    return 0;
}

// Whole-call drop-in: src/gpirtMCMC.cpp:5-117 behind src/RcppExports.cpp:16-30.
long long gpirt_debug_take_mcmc_trip(void);
static int g_last_mcmc_fallbacks = 0;
int gpirt_debug_last_mcmc_fallbacks(void) { return g_last_mcmc_fallbacks; }

int gpirt_mcmc(const double* h_y, int64_t n, int64_t m, const double* h_theta0, int S_it, int B_it,
               const double* h_pm, const double* h_ps, const double* h_step, const gpirt_options* opts,
               gpirt_rstream_t rs, gpirt_tick_fn tick, void* tick_ctx, double* h_theta_draws,
               double* h_beta_draws, double* h_f_draws, double* h_irfs)
{
    GP_ARG(h_y && h_theta0 && h_pm && h_ps && h_step && h_theta_draws && h_beta_draws && h_f_draws && h_irfs);
    GP_ARG(n > 0 && m > 0 && S_it >= 0 && B_it >= 0);
    gpirt_options o;
    if (opts) o = *opts; else gpirt_default_options(&o);
    gpirt_handle_t h = nullptr;
    GP_TRY(gpirt_create_own_stream(&h, o.device));
    { const long long trip = gpirt_debug_take_mcmc_trip(); if (trip > 0) h->trip_guard_at = trip; }
    gpirt_sampler_t s = nullptr;
    int rc = gpirt_sampler_create(&s, h, h_y, n, m, h_theta0, h_pm, h_ps, h_step, &o, rs);
    if (rc) { gpirt_destroy(h); return rc; }
    const int64_t N = s->N;
    const int total = S_it + B_it;
    const bool replay = stream_mode(s);
    std::vector<double> th((size_t)n);
    auto store_sync = [&](int slot) -> int {
        // theta_draws.row(slot), beta_draws.slice(slot), f_draws.slice(slot): :53-55, :99-101
        GP_TRY(gpirt_sampler_get(s, "theta", th.data(), n));
        for (int64_t i = 0; i < n; ++i) h_theta_draws[slot + i * (int64_t)(S_it + 1)] = th[(size_t)i];
        GP_TRY(gpirt_sampler_get(s, "beta", h_beta_draws + (int64_t)slot * 2 * m, 2 * m));
        GP_TRY(gpirt_sampler_get(s, "f", h_f_draws + (int64_t)slot * n * m, n * m));
        return 0;
    };
    rc = gpirt_sampler_init(s);
    if (!rc) rc = gpirt_sampler_check(s);
    if (!rc) rc = store_sync(0);

    if (replay) {
        // R-stream replay is item-sequential and drains the stream every iteration anyway (the cursor comes back to the
        // host): check, repair a hang-guard expiry in place (gpirt_sampler_check: nothing has read the new L yet) and store,
        // synchronously.
        for (int it = 0; it < total && !rc; ++it) {
            if (tick && tick(tick_ctx, it, total)) { set_error("interrupted"); rc = GPIRT_E_INTERRUPT; break; }
            rc = gpirt_sampler_step(s);
            if (!rc) rc = gpirt_sampler_check(s);
            if (!rc && it >= B_it) {
                rc = gpirt_sampler_accumulate_irf(s);                          // :103
                if (!rc) rc = store_sync(it - B_it + 1);
            }
        }
        if (!rc) rc = gpirt_sampler_finish_irfs(s, S_it, h_irfs);
        g_last_mcmc_fallbacks = h->guard_fallbacks;
        gpirt_sampler_destroy(s);
        gpirt_destroy(h);
        return rc;
    }

    // Item RNG: the host runs ahead of the device, errors are sticky words polled without draining the stream, and the
    // stored draws (src/gpirtMCMC.cpp:99-101; 64 MiB per stored iteration at the metric size) travel on a second stream
    // while the next iterations run.  Both hang off ONE mechanism: after every iteration the chain state is copied
    // device-to-device into one of three checkpoint slots (theta, beta, f, mu, mu*, f*, the IRF sums: ~1 % of an iteration),
    // followed by a read-back of the error words.  A checkpoint is VERIFIED once its error words have come back clean;
    // the host never enqueues iteration it before checkpoint it - 1 is verified (so it is at most two iterations ahead,
    // and a verified slot is never the one being overwritten).  Stored draws are copied out of verified checkpoints.
    // If the words come back with the panel kernel's hang guard raised (flagsync.h: its work-groups were not co-resident
    // within the spin bound -- a foreign tenant on the GPU can do that), the iterations enqueued since the last verified
    // checkpoint have consumed an unfinished factor: the state is rolled back to that checkpoint, its factor is rebuilt
    // from theta with the launch-per-step panel, the lost iterations are repeated on that panel too, and the chain goes
    // on -- the same draws as an undisturbed run (counter-based RNG keyed by the iteration; L equal to rounding).  Only a
    // second expiry without progress ends the call.
    constexpr int NS = 3;
    const size_t ck_doubles = (size_t)n + 2 * (size_t)m + 2 * (size_t)(n * m) + 3 * (size_t)(N * m);
    double* ck[NS] = { nullptr, nullptr, nullptr };
    hipEvent_t ev_flags[NS] = {}, ev_copied[NS] = {};
    bool copy_pending[NS] = { false, false, false };
    int copy_slot[NS] = { 0, 0, 0 };
    std::vector<double> th_stage[NS];
    hipStream_t copy_stream = nullptr;
    int* h_poll = nullptr;                                 // pinned, per slot: [potrf info + guard record (8) | flag0, flag1]
    auto fail_hip = [&](const char* what) { set_error("%s failed in gpirt_mcmc", what); return (int)GPIRT_E_HIP; };
    if (!rc) {
        bool ok = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) == hipSuccess &&
                  hipHostMalloc(&h_poll, NS * 16 * sizeof(int), hipHostMallocDefault) == hipSuccess;
        for (int q = 0; q < NS && ok; ++q) {
            ok = hipMalloc(&ck[q], ck_doubles * sizeof(double)) == hipSuccess &&
                 hipEventCreateWithFlags(&ev_flags[q], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&ev_copied[q], hipEventDisableTiming) == hipSuccess;
            th_stage[q].resize((size_t)n);
        }
        if (!ok) rc = fail_hip("allocation");
    }
    struct Part { double* p; size_t cnt; };
    auto parts = [&]() {
        return std::vector<Part>{ { s->theta, (size_t)n }, { s->beta, (size_t)(2 * m) }, { s->f, (size_t)(n * m) },
                                  { s->mu, (size_t)(n * m) }, { s->mu_star, (size_t)(N * m) }, { s->fstar, (size_t)(N * m) },
                                  { s->irf_sum, (size_t)(N * m) } };
    };
    auto save_ckpt = [&](int k) -> int {                    // state after k iterations -> slot k % NS, on the compute stream
        const int q = k % NS;
        hipStream_t st = h->stream;
        if (beta_sync(s) != 0) return fail_hip("draw_beta join");
        if (copy_pending[q] && hipStreamWaitEvent(st, ev_copied[q], 0) != hipSuccess) return fail_hip("hipStreamWaitEvent");
        double* d = ck[q];
        for (const Part& pt : parts()) {
            if (hipMemcpyAsync(d, pt.p, pt.cnt * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail_hip("checkpoint");
            d += pt.cnt;
        }
        if (hipMemcpyAsync(h_poll + 16 * q, h->d_info, 8 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(h_poll + 16 * q + 8, s->flags, 2 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipEventRecord(ev_flags[q], st) != hipSuccess)
            return fail_hip("flag read-back");
        return 0;
    };
    auto finish_store = [&](int q) -> int {                 // theta travels through a staging row: scatter it once it is here
        if (!copy_pending[q]) return 0;
        if (hipEventSynchronize(ev_copied[q]) != hipSuccess) return fail_hip("draw copy");
        const int slot = copy_slot[q];
        for (int64_t i = 0; i < n; ++i) h_theta_draws[slot + i * (int64_t)(S_it + 1)] = th_stage[q][(size_t)i];
        copy_pending[q] = false;
        return 0;
    };
    auto start_store = [&](int k) -> int {                  // verified checkpoint k -> draw slot k - B_it, on the copy stream
        const int q = k % NS, slot = k - B_it;
        GP_TRY(finish_store(q));
        const double* d = ck[q];
        if (hipMemcpyAsync(th_stage[q].data(), d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
            hipMemcpyAsync(h_beta_draws + (int64_t)slot * 2 * m, d + n, sizeof(double) * (size_t)(2 * m), hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
            hipMemcpyAsync(h_f_draws + (int64_t)slot * n * m, d + n + 2 * m, sizeof(double) * (size_t)(n * m), hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
            hipEventRecord(ev_copied[q], copy_stream) != hipSuccess)
            return fail_hip("draw copy");
        copy_pending[q] = true; copy_slot[q] = slot;
        return 0;
    };
    auto rollback = [&](int k) -> int {                     // chain state := verified checkpoint k, factor rebuilt on the fallback panel
        hipStream_t st = h->stream;
        if (hipStreamSynchronize(st) != hipSuccess) return fail_hip("hipStreamSynchronize");
        if (h->side) hipStreamSynchronize(h->side);
        if (s->haux) hipStreamSynchronize(s->haux->stream);
        s->beta_deferred = false; s->beta_pending = false; s->prep_pending = false; s->z_filled_iter = 0;
        const double* d = ck[k % NS];
        for (const Part& pt : parts()) {
            if (hipMemcpyAsync(pt.p, d, pt.cnt * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail_hip("rollback");
            d += pt.cnt;
        }
        if (hipMemsetAsync(s->flags, 0, 4 * sizeof(int), st) != hipSuccess) return fail_hip("rollback");
        s->iter = k;
        return recover_factor(s);
    };
    s->sticky_info = true;            // potrf no longer clears its info word: first failure sticks
    const int panel_mode = h->cfg.panel;
    int it = 0, verified = 0, fallback_until = 0, last_fallback = -1;
    if (!rc) rc = save_ckpt(0);
    if (!rc && hipEventSynchronize(ev_flags[0]) != hipSuccess) rc = fail_hip("hipEventSynchronize");
    while (!rc && verified < total) {
        if (it < total && verified >= it - 1) {
            if (tick && tick(tick_ctx, it, total)) { set_error("interrupted"); rc = GPIRT_E_INTERRUPT; break; }
            h->cfg.panel = (it < fallback_until) ? 2 : panel_mode;          // iterations lost to a guard expiry are repeated on the fallback panel
            if (h->aux) h->aux->cfg.panel = h->cfg.panel;
            rc = gpirt_sampler_step(s);
            h->cfg.panel = panel_mode;
            if (h->aux) h->aux->cfg.panel = panel_mode;
            if (!rc && it >= B_it) rc = gpirt_sampler_accumulate_irf(s);   // :103
            if (!rc) rc = save_ckpt(it + 1);
            ++it;
            continue;
        }
        const int k = verified + 1, q = k % NS;
        if (hipEventSynchronize(ev_flags[q]) != hipSuccess) { rc = fail_hip("hipEventSynchronize"); break; }
        const int* w = h_poll + 16 * q;
        if (w[1] != 0) {
            if (panel_mode == 2 || last_fallback == verified) { rc = report_panel_guard(h, w, h->stream); break; }
            rc = rollback(verified);
            last_fallback = verified; fallback_until = it; it = verified;
            continue;
        }
        if (w[0] > 0) {
            set_error("chol(): decomposition failed (leading minor of order %d is not positive definite)", w[0]);
            rc = w[0];
        } else if (w[8] != 0) {
            set_error("sampler state is not finite (flag %d)", w[8]); rc = w[8];
        } else if (w[9] != 0) {
            rc = report_degenerate_theta(s, w[9]);
        } else {
            verified = k;
            if (k > B_it) rc = start_store(k);
        }
    }
    for (int q = 0; q < NS; ++q) { const int r2 = finish_store(q); if (!rc) rc = r2; }
    if (rc != GPIRT_E_INTERRUPT) {
        const int rc2 = gpirt_sampler_check(s);                            // final, synchronising
        if (!rc) rc = rc2;
    }
    if (copy_stream) { hipStreamSynchronize(copy_stream); hipStreamDestroy(copy_stream); }
    for (int q = 0; q < NS; ++q) {
        if (ck[q]) hipFree(ck[q]);
        if (ev_flags[q]) hipEventDestroy(ev_flags[q]);
        if (ev_copied[q]) hipEventDestroy(ev_copied[q]);
    }
    if (h_poll) hipHostFree(h_poll);
    if (!rc) rc = gpirt_sampler_finish_irfs(s, S_it, h_irfs);
    g_last_mcmc_fallbacks = h->guard_fallbacks;
    gpirt_sampler_destroy(s);
    gpirt_destroy(h);
    return rc;
}

}  // extern "C"
