// common.h -- shared host/device helpers of libgpirt_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "../../include/gpirt_hip.h"

namespace gpirt {

// ---------------------------------------------------------------- errors -------------------
void set_error(const char* fmt, ...);

#define GP_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            gpirt::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                             __LINE__);                                                      \
            return GPIRT_E_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define GP_ARG(cond)                                                                         \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            gpirt::set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);         \
            return GPIRT_E_ARG;                                                              \
        }                                                                                    \
    } while (0)

#define GP_TRY(call)                                                                         \
    do {                                                                                     \
        int r_ = (call);                                                                     \
        if (r_ != 0) return r_;                                                              \
    } while (0)

// ---------------------------------------------------------------- handle -------------------
// Event-pair profiler for the syrk launches of the factorisation (the MFMA work of arma::chol): pairs are recorded
// on the launch stream without synchronising and resolved later by gpirt_prof_syrk().  Classes:
//   0  trailing update, 128-tile kernel   1  trailing update, 64-tile kernel   2  update inside an outer panel (K = the first sub-panel's width)
// ... and, with the same instrument, two kernels of draw_f:
//   3  nu = L Z, the triangular product of the item-keyed draw_f (src/mvnormal.h:10 for all m columns)
//   4  rs3_products_kernel, the pass over L of the R-stream replay's draw_f (bytes: the lower triangle of L)
// ... and on draw_theta's product:
//   5  tf_mfma_kernel, the log-posterior product in fixed point (theta_fixed.hip; "flops" = int8 multiply-adds x 2 of the
//      seven digit planes, 7 x 2 x 1001 x n x 2m)
constexpr int PROF_CLASSES = 6;
struct ProfPair { hipEvent_t e0, e1; double flops; int cls; double bytes; };
struct Prof {
    bool        enabled = false;
    double      ms[PROF_CLASSES] = {};
    int64_t     launches[PROF_CLASSES] = {};
    double      flops[PROF_CLASSES] = {};
    double      bytes[PROF_CLASSES] = {};     // algorithmic: C trapezoid read + written once, the panel operand read once
    std::vector<ProfPair> pending;
    std::vector<ProfPair> free_pairs;
};

// The library's environment switches.  Read ONCE per process (env_config(), api.hip), copied into every handle at
// creation; the non-geometry ones can be changed per handle through gpirt_config_set (tests do, instead of editing the
// environment between calls).  Defaults are the measured optimum at n = 8192 (README.md has the table).
struct Config {
    int  nbo = 1024, nbp = 0;     // GPIRT_NBO / GPIRT_NBP: outer panel width (K of a trailing update) / sub-panel width (0: by size, potrf_subpanel_width)
    int  lookahead = 1;           // GPIRT_LOOKAHEAD: 1 = the next panel is factored on a side stream beside the updates, 2 = off
    int  panel = 1;               // GPIRT_PANEL: 1 = persistent sub-panel kernel (panel.hip), 2 = launch-per-step panel (the
                                  //   fallback a hang-guard expiry refactors with)
    int  defer = 0;               // GPIRT_DEFER: 3 = trailing updates deferred block column by block column, 2 = plain
                                  //   right-looking order, 0 = by size (3 up to n = 14336)
    int  trsm_inv = 1;            // GPIRT_TRSM_INV: 2 = every trsm leaf is a substitution (parity attribution)
    int  ll_exact = 0;            // GPIRT_LL_EXACT: 1 = the slice kernel evaluates log(1 + exp(-a)) as written
    int  theta_fixed = 1;         // GPIRT_THETA_FIXED: 2 = draw_theta's log-posterior product as an fp64 GEMM (theta_fixed.hip)
    int  ess_screen = 1;          // GPIRT_ESS_SCREEN: 2 = every trial point of the slice kernel is evaluated in full precision
    int  bordered = 1;            // GPIRT_BORDERED: 2 = draw_fstar solves for L^-1 K(theta, c) / L^-1 k* explicitly
    int  early_inv = 1;           // GPIRT_EARLY_INV: 2 = no side work beside the factorisation's last outer panel
    int  prep_early = 1;          // GPIRT_PREP_EARLY: 2 = the factor-only part of the rank-r draw_fstar waits for nu = L z
    int  rs_lr = 1;               // GPIRT_RS_LR: 2 = the predictor's every pass reads all of L as floats (no structured form, rs_lr.hip)
    int  rs_predict = 1;          // GPIRT_RS_PREDICT: 2 = the R-stream replay's draw_f runs every pass over L in fp64 (one phase, rng_ess.hip)
    int  guard_verbose = 0;       // GPIRT_GUARD_VERBOSE: 1 = a hang-guard fallback prints the guard record to stderr
};
const Config& env_config();

}  // namespace gpirt

struct gpirt_handle_s {
    int          device = 0;
    gpirt::Config cfg;
    hipStream_t  stream = nullptr;
    bool         own_stream = false;
    // small persistent workspace
    int*         d_info = nullptr;       // potrf info (device)
    int*         h_info = nullptr;       // pinned mirror
    double*      d_work = nullptr;       // generic scratch (grown on demand)
    size_t       work_bytes = 0;
    gpirt::Prof  prof;
    // look-ahead Cholesky: high-priority side stream for the panel chain + fork/join events
    hipStream_t  side = nullptr;
    hipEvent_t   ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr, ev_half = nullptr;
    // The samplers' side work (the factor-only part of draw_fstar, block inverses, the next draw_f's normals, draw_beta) runs
    // on ONE high-priority handle per main handle, shared by every sampler created on it (samplers of a handle are driven
    // one after the other anyway).  Round 3: a handle per SAMPLER meant a new high-priority stream per sampler, and with
    // three samplers alive on one handle every stage of the third ran 1.3-1.9x slower (tools/two_samplers_probe.py).
    gpirt_handle_t aux = nullptr;
    const double* inv_partial_L = nullptr;   // (on the aux handle) the factor whose block inverses are PARTLY built ...
    int64_t      inv_partial_pairs = 0;      //   ... 512-block pairs [0, inv_partial_pairs)
    bool         slim_leaf = false;       // trsm_inverses_build launches the 256-register form of its leaf kernel (trsm.hip)
    hipEvent_t   ev_prelast = nullptr;    // fires when every outer panel but the last is final (columns [0, prelast_cols))
    int64_t      prelast_cols = 0;        //   ... of the factorisation enqueued last (0: no such point, e.g. a single panel)
    // persistent panel kernel (panel.hip): one progress counter per 64-row block, epoch-tagged
    unsigned long long* d_prog = nullptr;
    double*      d_winv = nullptr;        // panel.hip: inverses of the diagonal blocks' 16 x 16 blocks, handed along the pivot chain
    int64_t      winv_blocks = 0;         //   ... one 4 x 256 slot per 64-column block of the matrix
    double*      d_defer_ws = nullptr;    // potrf.hip: slabs of the panel-parallel deferred updates (launch_syrk_panels)
    size_t       defer_ws_bytes = 0;
    size_t       prog_cap = 0;
    unsigned long long prog_seq = 0;
    long long*   panel_trace = nullptr;   // debug stamps (micro-benchmarks; gpirt_debug_panel_trace)
    int64_t      panel_trace_k0 = -1;     // >= 0: only the sub-panel launch that starts at this column stamps
    int64_t      panel_trace_cap = 0;     // entries allocated by gpirt_debug_panel_trace
    int          n_cu = 0;                // compute units of `device` (grid cap of the persistent kernel)
    // hang-guard fallback (sampler.hip recover_factor, api.hip finish_info): a factorisation whose persistent kernel gave
    // up on a progress counter is repeated once with the launch-per-step panel; counted here (gpirt_guard_fallbacks)
    // Samplers borrow the handle (its stream, workspaces, side handle).  A handle destroyed while samplers still live --
    // e.g. a host language tearing objects down in arbitrary order at exit -- is only marked (zombie) and freed by the last
    // sampler's destroy: round 3's `std::bad_variant_access` abort at interpreter exit was gpirt_sampler_destroy draining the
    // stream of a handle that had already been freed (DESIGN_HISTORY.md section 8.1).
    int          live_samplers = 0;
    bool         zombie = false;
    int          guard_fallbacks = 0;
    int64_t      cur_nbp = 512;           // sub-panel width of the factorisation being enqueued (potrf_subpanel_width(n))
    long long    factor_count = 0;        // factorisations enqueued on this handle (launch_potrf_lower)
    int          rs_trace_pass = -1;      // debug (gpirt_debug_rs_trace): the pass of every replayed draw_f whose kernels stamp their phases
    int          rs_mispredict = 0;       // debug (gpirt_debug_rs_mispredict): the replay's predictor is off by one at every n-th item
    int          rs_cand_limit = 0;       // debug (gpirt_debug_rs_cand_limit): candidates the replay's draw_f may use (0: all)
    long long    trip_guard_at = -1;      // debug (gpirt_debug_trip_guard): the factorisation with this count raises the
                                          //   guard word and poisons its result behind itself, as an expiry would leave it
    bool         panel_attr_set = false;  // dynamic-LDS attribute of panel_ll_kernel set on this device
    // gemm_f64.hip: parts of automatically split-K products
    double*      d_splitk = nullptr;
    size_t       splitk_bytes = 0;
    // trsm.hip: inverses of L's 256 x 256 diagonal blocks (rebuilt per call) + a 256 x nrhs product buffer
    double*      d_trsm_winv = nullptr;
    size_t       trsm_winv_bytes = 0;
    double*      d_trsm_tmp = nullptr;
    size_t       trsm_tmp_bytes = 0;
    double*      d_trsm_wquad = nullptr;  // 1024 x 1024 inverses for thin solves (built with the 512 ones)
    size_t       trsm_wquad_bytes = 0;
    int64_t      trsm_quads = 0;
    const double* trsm_winv_L = nullptr;  // the factor the inverses belong to (launch_trsm_lower, reuse_inverses)
    int64_t      trsm_winv_n = 0, trsm_winv_ld = 0;
};

namespace gpirt {

int  ensure_work(gpirt_handle_t h, size_t bytes);   // grows h->d_work (syncs when it reallocates)

// ---------------------------------------------------------------- device math --------------
typedef double d4 __attribute__((ext_vector_type(4)));

#define GP_2PI 6.283185307179586476925286766559

// Philox4x32-10 (same contract as the oracle; GPIRT_RNG_ITEM)
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one uniform in (0,1) per (seed, iteration, stage, item, index): 52 random bits + half an ulp
__host__ __device__ inline double item_uniform(uint64_t seed, uint32_t iter, uint32_t stage,
                                               uint32_t item, uint32_t index)
{
    uint32_t o[4];
    philox4x32_10(index, item, stage, iter, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    uint64_t v = ((uint64_t)(o[0] >> 6) << 26) | (uint64_t)(o[1] >> 6);
    return ((double)v + 0.5) * 2.220446049250313e-16;
}

// R's qnorm(p, 0, 1, lower, !log): Wichura AS241 PPND16, evaluated in R's Horner order
__host__ __device__ inline double qnorm_as241(double p)
{
    double q = p - 0.5, r, val;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r
                        + 67265.770927008700853) * r + 45921.953931549871457) * r
                      + 13731.693765509461125) * r + 1971.5909503065514427) * r
                    + 133.14166789178437745) * r + 3.387132872796366608)
              / (((((((r * 5226.495278852545925 + 28729.085735721942674) * r
                      + 39307.89580009271061) * r + 21213.794301586595867) * r
                    + 5394.1960214247511077) * r + 687.1870074920579083) * r
                  + 42.313330701600911252) * r + 1.0);
        return val;
    }
    r = (q < 0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    if (r <= 5.0) {
        r += -1.6;
        val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r
                    + 0.24178072517745061177) * r + 1.27045825245236838258) * r
                  + 3.64784832476320460504) * r + 5.7694972214606914055) * r
                + 4.6303378461565452959) * r + 1.42343711074968357734)
              / (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r
                      + 0.0151986665636164571966) * r + 0.14810397642748007459) * r
                    + 0.68976733498510000455) * r + 1.6763848301838038494) * r
                  + 2.05319162663775882187) * r + 1.0);
    } else {
        r += -5.0;
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r
                    + 0.0012426609473880784386) * r + 0.026532189526576123093) * r
                  + 0.29656057182850489123) * r + 1.7848265399172913358) * r
                + 5.4637849111641143699) * r + 6.6579046435011037772)
              / (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r
                      + 1.8463183175100546818e-5) * r + 7.868691311456132591e-4) * r
                    + 0.0148753612908506148525) * r + 0.13692988092273580531) * r
                  + 0.59983220655588793769) * r + 1.0);
    }
    if (q < 0.0) val = -val;
    return val;
}

// R's norm_rand() under INVERSION from two consecutive stream uniforms
__host__ __device__ inline double rnorm_from_two(double u1, double u2)
{
    const double BIG = 134217728.0;
    double u = (double)(int)(BIG * u1) + u2;
    return qnorm_as241(u / BIG);
}

// one term of ll()/ll_bar(): log(1 + exp(-a)), verbatim (src/log-likelihood.cpp:20,34)
__device__ inline double ll_term(double a) { return log(1 + exp(-a)); }

// block-wide sum for 256-thread blocks, fixed reduction tree (deterministic)
__device__ inline double block_sum_256(double v, double* smem /* >= 4 doubles */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    double r = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    return r;
}

// the same for 512-thread blocks (8 wavefronts)
__device__ inline double block_sum_512(double v, double* smem /* >= 8 doubles */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    return ((smem[0] + smem[1]) + (smem[2] + smem[3])) + ((smem[4] + smem[5]) + (smem[6] + smem[7]));
}

}  // namespace gpirt
