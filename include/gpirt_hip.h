/*
 * gpirt_hip.h -- C ABI of libgpirt_hip.so: the MI355X (gfx950) implementation of the per-iteration
 * GP linear algebra of duckmayr/gpirt's gpirtMCMC(), behind the reference's own boundary.
 *
 * Boundary being replaced (paths relative to the upstream repository):
 *   R/RcppExports.R:4-6      .Call(`_gpirt_gpirtMCMC`, y, theta, S, B, prior_means, prior_sds, steps)
 *   src/RcppExports.cpp:16-30  SEXP _gpirt_gpirtMCMC(SEXP x 7)  (Rcpp glue, RNGScope, BEGIN/END_RCPP)
 *   src/gpirtMCMC.cpp:5-9    Rcpp::List gpirtMCMC(const arma::mat& y, arma::vec theta, int, int, ...)
 *   src/gpirt.h:4-28         internal prototypes K / draw_f / draw_fstar / draw_theta / draw_beta / ll_bar
 *
 * Conventions: plain pointers and sizes only; every matrix is column-major fp64 (Armadillo / R
 * layout); y holds -1 / +1 / NaN; int64_t sizes; `void* stream` is a hipStream_t (NULL = the
 * handle's stream).  Pointers named d_* are DEVICE pointers, h_* are HOST pointers.
 * Every function returns 0 on success; >0 = LAPACK-style potrf info (order of the leading minor
 * that is not positive definite -- what makes arma::chol throw "decomposition failed",
 * src/gpirtMCMC.cpp:17); <0 = GPIRT_E_* .  gpirt_last_error() describes the last failure of the
 * calling thread.  No function throws across the boundary.
 *
 * The library has NO CPU fallback: without a usable gfx950 device every compute entry fails with
 * GPIRT_E_NODEVICE.
 */
#ifndef GPIRT_HIP_H
#define GPIRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPIRT_NGRID 1001           /* theta_star = regspace(-5, 0.01, 5): src/gpirtMCMC.cpp:35 */
#define GPIRT_JITTER 0.001         /* S.diag() += 0.001: src/gpirtMCMC.cpp:16,77,96 */

enum {
    GPIRT_OK          = 0,
    GPIRT_E_ARG       = -1,   /* bad argument (NULL pointer, negative size, ...) */
    GPIRT_E_HIP       = -2,   /* a HIP runtime call failed */
    GPIRT_E_NODEVICE  = -3,   /* no gfx950 device visible */
    GPIRT_E_ALLOC     = -4,   /* device or host allocation failed */
    GPIRT_E_RNG       = -5,   /* R-stream replay ran out of pre-generated uniforms */
    GPIRT_E_INTERRUPT = -6,   /* the tick callback asked to stop */
    GPIRT_E_NUMERIC   = -7    /* non-finite state (e.g. ESS did not terminate) */
};

/* RNG contracts:
 *  GPIRT_RNG_RSTREAM  exact replay of R's Mersenne-Twister/inversion stream in the reference's
 *                     consumption order (item-sequential draw_f);
 *  GPIRT_RNG_ITEM     counter-based Philox4x32-10 sub-streams keyed by (seed, iteration, stage,
 *                     item): same algorithm, items independent, batched trmm legal. */
enum { GPIRT_RNG_RSTREAM = 0, GPIRT_RNG_ITEM = 1 };

/* stage ids of the GPIRT_RNG_ITEM contract */
enum {
    GPIRT_ST_INIT_F = 1, GPIRT_ST_INIT_BETA = 2, GPIRT_ST_F_Z = 3, GPIRT_ST_F_ESS = 4,
    GPIRT_ST_FSTAR = 5, GPIRT_ST_THETA = 6, GPIRT_ST_BETA = 7
};

typedef struct gpirt_handle_s*  gpirt_handle_t;
typedef struct gpirt_sampler_s* gpirt_sampler_t;

/* ---------------------------------------------------------------- library / handle ------ */
int         gpirt_version(void);    /* 102: gpirt_potrf_subpanel_width(n) takes the order of the matrix; 101: named gpirt_options fields, gpirt_fast_options */
const char* gpirt_last_error(void);
int         gpirt_device_count(int* count);
/* device < 0: current device.  stream is a hipStream_t; NULL is HIP's default (null) stream. */
int         gpirt_create(gpirt_handle_t* h, int device, void* stream);
/* same, but the handle creates and owns a non-blocking stream of its own */
int         gpirt_create_own_stream(gpirt_handle_t* h, int device);
int         gpirt_destroy(gpirt_handle_t h);
int         gpirt_synchronize(gpirt_handle_t h);
int         gpirt_set_stream(gpirt_handle_t h, void* stream);
/* The library's switches (README.md: GPIRT_PANEL, GPIRT_LOOKAHEAD, GPIRT_DEFER, GPIRT_TRSM_INV, GPIRT_LL_EXACT,
 * GPIRT_BORDERED, GPIRT_EARLY_INV, GPIRT_PREP_EARLY; GPIRT_NBO / GPIRT_NBP read-only).  The environment
 * is read ONCE per process; every handle starts from those values and a caller may change them per handle here (drains
 * the handle's stream; takes effect from the next call on).  Nothing in the library reads the environment per call. */
int         gpirt_config_get(gpirt_handle_t h, const char* name, int* value);
int         gpirt_config_set(gpirt_handle_t h, const char* name, int value);
/* Hang-guard fallbacks taken on this handle so far: a factorisation whose persistent sub-panel kernel gave up on a
 * progress counter (its work-groups were not co-resident within the spin bound, e.g. beside a foreign tenant of the GPU)
 * is repeated ONCE from the intact theta with the launch-per-step panel (GPIRT_PANEL=2's path) by gpirt_factor,
 * gpirt_sampler_check and gpirt_mcmc; only a failure of the repeat is an error. */
int         gpirt_guard_fallbacks(gpirt_handle_t h, int* count);
/* Debug: the nth factorisation enqueued on h from now (nth >= 1; 0 disarms) ends the way a guard expiry leaves it -- guard
 * word raised, result unfinished -- without spinning any kernel.  h == NULL arms the handle the NEXT gpirt_mcmc call
 * creates for itself (that call reports its fallbacks through gpirt_debug_last_mcmc_fallbacks). */
int         gpirt_debug_trip_guard(gpirt_handle_t h, int nth);
/* Tests only: a pass of the R-stream replay's draw_f resolves up to three items: the anchor, the next
 * item at one of 15 places its predecessor's uniform count selects, the one after at one of 16.  limit > 0 clamps both
 * candidate counts to min(15 | 16, limit) (0: the full 15 / 16), so that passes which end early -- a predecessor's slice loop
 * consumed more than the slot holds candidates for, the next pass starts at the first unresolved item -- are exercised at
 * will.  There is no host fallback: the anchor is device-side state. */
int         gpirt_debug_rs_cand_limit(gpirt_handle_t h, int limit);
/* Tests only: the R-stream replay's draw_f predicts every item's start in R's stream on a single-precision copy of L and
 * verifies all items exactly afterwards; every > 0 makes the predictor wrong on purpose at every
 * every-th item, so the verification's discard-and-resume path runs.  The draws must not change. */
int         gpirt_debug_rs_mispredict(gpirt_handle_t h, int every);
/* Debug: pass number `pass` (0-based; < 0: none) of every R-stream draw_f on this handle leaves the in-kernel time stamps of
 * its two kernels in the sampler's "rs_trace" array (gpirt_sampler_get, 128 64-bit words, 100 MHz): tools/rs_trace.py. */
int         gpirt_debug_rs_trace(gpirt_handle_t h, int pass);
int         gpirt_debug_last_mcmc_fallbacks(void);
/* Peak fp64 MFMA rate of this device measured by a back-to-back v_mfma_f64_16x16x4_f64 loop
 * (TFLOP/s); used to calibrate the roofline. */
int         gpirt_calibrate_mfma_f64(gpirt_handle_t h, double* tflops);

/* ---------------------------------------------------------------- operators ------------- */
/* K(): src/covariance-function.cpp:3-14.  d_out (n1 x n2, leading dimension ld) =
 * exp(-0.5 (x1_i - x2_j)^2); `jitter` is added where i == j (src/gpirtMCMC.cpp:16; pass 0 for
 * K(theta, theta_star), src/draw-fstar.cpp:17). */
int gpirt_se_kernel(gpirt_handle_t h, const double* d_x1, int64_t n1, const double* d_x2,
                    int64_t n2, double* d_out, int64_t ld, double jitter);

/* arma::chol(S, "lower"): src/gpirtMCMC.cpp:17,78,97.  In place on d_A (n x n, lda): on exit the
 * lower triangle holds L and the strict upper triangle is zero.  Blocked right-looking
 * factorisation, trailing update on fp64 MFMA.  Returns info (>0) if S is not positive definite. */
int gpirt_potrf_lower(gpirt_handle_t h, double* d_A, int64_t n, int64_t lda);

/* Fused K(theta,theta) + jitter + chol: src/gpirtMCMC.cpp:15-17,76-78,95-97.  d_L n x n. */
int gpirt_factor(gpirt_handle_t h, const double* d_theta, int64_t n, double* d_L, int64_t ldl);

/* The same factorisation in pieces, for a host that distributes it over several GPUs (one process each): outer panel
 * p = columns [p W, min((p + 1) W, n)), W = gpirt_potrf_panel_width().  With 1-D block-cyclic ownership of the outer
 * panels the owner of p calls panel_factor once its columns carry the updates of every panel q < p, the finished
 * panel travels to the other ranks through panel_copy (rows [pW, n) of the panel <-> a dense (n - pW) x w buffer the
 * host broadcasts), and every rank applies it to the block columns c > p it owns with panel_update.  The pieces are
 * the launches gpirt_potrf_lower itself makes, so the assembled factor is bit-identical to it.  Nothing here
 * synchronises; gpirt_potrf_begin clears the info word and gpirt_potrf_finish drains the stream and returns it. */
int64_t gpirt_potrf_panel_width(void);
int gpirt_potrf_begin(gpirt_handle_t h);
int gpirt_potrf_panel_factor(gpirt_handle_t h, double* d_A, int64_t n, int64_t lda, int64_t p);
int gpirt_potrf_panel_update(gpirt_handle_t h, double* d_A, int64_t n, int64_t lda, int64_t p, int64_t c);
int gpirt_potrf_panel_copy(gpirt_handle_t h, double* d_A, int64_t n, int64_t lda, int64_t p, double* d_buf, int to_buf);
int gpirt_potrf_finish(gpirt_handle_t h);
/* The same pieces by HALVES of an outer panel, for a host that pipelines its broadcasts: an outer panel is factored as a
 * first sub-panel of gpirt_potrf_subpanel_width(n) columns (GPIRT_NBP, or chosen by the order n of the matrix) and the rest, and the next panel's first columns take the panel's
 * update as two products (first sub-panel, then the rest -- the same rule launch_potrf_lower follows on one GPU), so
 *   half: 0 = the panel's first sub-panel, 1 = the rest of it, 2 = the whole panel        (factor / copy)
 *   part: 0 = what needs only panel p's first sub-panel, 1 = everything else, 2 = all     (update)
 * let the next owner start on a panel's first half while its second half is still being factored or travelling.
 * Factoring / updating by halves launches exactly what the whole-panel calls launch, in the same order: L is the same
 * bit for bit.  copy_part moves the part's columns, rows from the part's first row down (dense, ld = that row count);
 * buf_doubles is the capacity of d_buf in doubles: a part that does not fit is refused (GPIRT_E_ARG), never truncated
 * (half 1 is W - H columns wide, wider than half 0 whenever GPIRT_NBP < GPIRT_NBO / 2). */
int64_t gpirt_potrf_subpanel_width(int64_t n);
int gpirt_potrf_panel_factor_part(gpirt_handle_t h, double* d_A, int64_t n, int64_t lda, int64_t p, int half);
int gpirt_potrf_panel_update_part(gpirt_handle_t h, double* d_A, int64_t n, int64_t lda, int64_t p, int64_t c, int part);
int gpirt_potrf_panel_copy_part(gpirt_handle_t h, double* d_A, int64_t n, int64_t lda, int64_t p, int half, double* d_buf,
                                int64_t buf_doubles, int to_buf);
/* Debug aid for hosts that enqueue collectives between the pieces: *busy = bit mask of the handle's INTERNAL streams that
 * still have work in flight (0 = every piece has joined the handle's stream, as each must before it returns). */
int gpirt_debug_streams_busy(gpirt_handle_t h, int* busy);
/* One term of ll() (src/log-likelihood.cpp:20,34), log(1 + exp(-a)), elementwise for the n device values d_a:
 * fast = 0 the formula as written through the device library's exp and log (ll_bar, draw_beta, draw_theta and every
 * R-stream replay use it), fast = 1 the form of csrc/ll_fast.h that the elliptical-slice kernel of the item-keyed RNG
 * evaluates (within 3 ulp of the exact value; GPIRT_LL_EXACT=1 makes that kernel use the written form too), fast = 2 the
 * single-precision SCREEN that kernel tries first at every trial point: it only decides an accept test whose sum is
 * further from the slice level than the screen's error bound (4e-6 per row), everything closer is repeated in full
 * precision, so the decisions and the draws are the full-precision ones (GPIRT_ESS_SCREEN=2 switches the screen off). */
/* In-kernel time stamps of ONE sub-panel launch of the factorisation as it runs inside the schedule (100 MHz wall clock:
 * [row block relative to the launch][step 0..39][slot 0..7], the layout tools/micro/panel_bench.hip prints).
 * host_out == NULL arms it for the launch that starts at column k0 (count = entries to allocate, >= 40 * 8 * row blocks;
 * count = 0 disarms and frees); otherwise copies `count` entries out.  tools/panel_trace_insitu.py. */
int gpirt_debug_panel_trace(gpirt_handle_t h, int64_t k0, long long* host_out, int64_t count);
int gpirt_debug_ll_term(gpirt_handle_t h, const double* d_a, int64_t n, double* d_out, int fast);
/* The log-posterior of draw_theta before the prior (GPIRT_NGRID x n, column i = respondent i): the product
 * sum_j [y_ij = +-1] -log(1 + exp(-+ f*_gj)) as gpirt_draw_theta and the sampler form it -- in exact fixed point on the int8
 * matrix cores (csrc/theta_fixed.hip; every term rounded once to 54 bits of its grid row's range, the sums exact), or with
 * GPIRT_THETA_FIXED=2 as the fp64 GEMM.  *fell_back = 1 when the fixed-point form handed over to the fp64 product on its
 * own (|f*| > 709 or non-finite somewhere: the formula as written overflows there).  Synchronises the handle's stream. */
int gpirt_debug_theta_logpost(gpirt_handle_t h, const double* d_y, const double* d_fstar, int64_t n, int64_t m,
                              double* d_logpost_out, int* fell_back);
/* The same product with in-kernel stamps: six 64-bit words per work-group of the int8 kernel (the first 1024 of them) --
 * shader clock (s_memtime) and 100 MHz wall clock (s_memrealtime) at its start, after its main loop and at its end; what the
 * clock the chip holds under this kernel and the share of its prologue and epilogue are read from (tools/theta_clock.py). */
int gpirt_debug_theta_clock(gpirt_handle_t h, const double* d_y, const double* d_fstar, int64_t n, int64_t m,
                            long long* host_stamps, int64_t count);

/* rmvnorm()'s product `cholS * res` (src/mvnormal.h:10) for all item columns at once:
 * d_out (n x m) = L * Z with L lower triangular; the strict upper triangle of d_L must hold zeros
 * (as gpirt_potrf_lower / arma::chol leave it). */
int gpirt_trmm_lz(gpirt_handle_t h, const double* d_L, int64_t n, int64_t ldl, const double* d_Z,
                  int64_t m, int64_t ldz, double* d_out, int64_t ldo);

/* solve(trimatl(L), B) (trans = 0) and solve(trimatu(L.t()), B) (trans = 1):
 * src/draw-fstar.cpp:7,19.  In place on d_B (n x nrhs, ldb). */
int gpirt_trsm_lower(gpirt_handle_t h, const double* d_L, int64_t n, int64_t ldl, double* d_B,
                     int64_t nrhs, int64_t ldb, int trans);

/* General fp64 MFMA GEMM used by every stage: C = alpha * op(A) op(B) + beta * C.
 * ta/tb: 0 = as stored, 1 = transposed.  C is M x N. */
int gpirt_gemm(gpirt_handle_t h, int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha,
               const double* d_A, int64_t lda, const double* d_B, int64_t ldb, double beta,
               double* d_C, int64_t ldc);

/* ll_bar() for every column: src/log-likelihood.cpp:25-37.  d_out[j] = ll_bar(f_j, y_j, mu_j).
 * d_mu may be NULL (then this is ll(), :12-23). */
int gpirt_ll_bar(gpirt_handle_t h, const double* d_f, const double* d_y, const double* d_mu,
                 int64_t n, int64_t m, double* d_out);

/* draw_f(): src/draw-f.cpp:64-73 under GPIRT_RNG_ITEM: Z ~ N(0,1) from (seed, iter) sub-streams,
 * nu = L Z as one trmm, then one elliptical-slice update per column (ess(), :21-60).
 * In place on d_f (n x m).  d_k_out (m ints, may be NULL) receives the rejection counts. */
int gpirt_draw_f(gpirt_handle_t h, double* d_f, const double* d_y, const double* d_L, int64_t ldl,
                 const double* d_mu, int64_t n, int64_t m, uint64_t seed, uint32_t iter,
                 int* d_k_out);

/* draw_fstar(): src/draw-fstar.cpp:10-31 under GPIRT_RNG_ITEM.  d_out (N x m), N = GPIRT_NGRID.
 * fused = 0: alpha = L^-T L^-1 f as in the reference (double_solve, :3-8);
 * fused = 1: mean = (L^-1 kstar)^T (L^-1 f), algebraically identical, one trsm fewer.
 * d_s_out (N) and d_mean_out (N x m) may be NULL. */
int gpirt_draw_fstar(gpirt_handle_t h, const double* d_f, const double* d_theta, const double* d_L,
                     int64_t ldl, const double* d_mu_star, int64_t n, int64_t m, uint64_t seed,
                     uint32_t iter, int fused, double* d_out, double* d_s_out, double* d_mean_out);

/* draw_theta(): src/draw-theta.cpp:3-37 under GPIRT_RNG_ITEM, as one fp64 MFMA GEMM over the
 * +1 / -1 indicator matrices of y.  stabilise = 1 subtracts the column maximum before exp. */
int gpirt_draw_theta(gpirt_handle_t h, const double* d_y, const double* d_fstar, int64_t n,
                     int64_t m, uint64_t seed, uint32_t iter, int stabilise, double* d_theta_out,
                     int* d_degenerate);

/* draw_beta(): src/draw-beta.cpp:3-41 under GPIRT_RNG_ITEM.  In place on d_beta (2 x m). */
int gpirt_draw_beta(gpirt_handle_t h, double* d_beta, const double* d_theta, const double* d_y,
                    const double* d_f, const double* d_prior_means, const double* d_prior_sds,
                    const double* d_step_sizes, int64_t n, int64_t m, uint64_t seed, uint32_t iter);

/* The item-RNG primitives, exported so hosts and tests can address the same sub-streams. */
int gpirt_item_uniforms(gpirt_handle_t h, uint64_t seed, uint32_t iter, uint32_t stage,
                        uint32_t item0, int64_t n_items, int64_t n_index, double* d_out);
int gpirt_item_normals(gpirt_handle_t h, uint64_t seed, uint32_t iter, uint32_t stage,
                       uint32_t item0, int64_t n_items, int64_t n_index, double* d_out);

/* ---------------------------------------------------------------- R's RNG on the host --- */
/* set.seed(seed) + unif_rand()/norm_rand() of R's default Mersenne-Twister/inversion generator.
 * Host code (R does this itself before the call: theta_init <- rnorm(n), R/gpirtMCMC.R:95-97). */
typedef struct gpirt_rstream_s* gpirt_rstream_t;
int gpirt_rstream_create(gpirt_rstream_t* r, uint32_t seed);
int gpirt_rstream_from_state(gpirt_rstream_t* r, const uint32_t mt[624], int mti);
int gpirt_rstream_get_state(gpirt_rstream_t r, uint32_t mt[624], int* mti);
int gpirt_rstream_destroy(gpirt_rstream_t r);
int gpirt_rstream_unif(gpirt_rstream_t r, double* h_out, int64_t n);
int gpirt_rstream_norm(gpirt_rstream_t r, double* h_out, int64_t n);

/* ---------------------------------------------------------------- sampler --------------- */
typedef int (*gpirt_tick_fn)(void* ctx, int iter, int total);  /* nonzero return = stop */

typedef struct gpirt_options {
    int      rng_kind;        /* GPIRT_RNG_RSTREAM or GPIRT_RNG_ITEM */
    uint64_t seed;            /* GPIRT_RNG_ITEM key */
    int      theta_stabilise; /* 1 = subtract the row maximum before exp in draw_theta */
    int      fstar_fused;     /* see gpirt_draw_fstar */
    int      device;          /* < 0: current device */
    int      reserved0;       /* must be 0 (round 1 reserved this slot for a hipGraph replay switch that was never
                               * built: the host enqueues one iteration in 0.40 ms against 8.3 ms on the device,
                               * tools/host_time_probe.py, so a graph has nothing to win; the slot keeps the layout) */
    /* item sharding (one process per GPU): this rank owns item columns [item0, item0 + m) of a
     * global problem with m_total items; y / priors / outputs passed in are the LOCAL columns. */
    int64_t  item0;
    int64_t  m_total;
    int      reserved1;       /* must be 0 */
    int      kernel_fp32;     /* 1: K(theta, theta) is built with single-precision exp() before the fp64 factorisation
                               * (BASELINE config C5; parity then only statistical) */
    int      kstar_rank;      /* r (16..128, multiple of 16; needs fstar_fused): draw_fstar works with the rank-r Chebyshev
                               * factorisation K(theta, theta*) = K(theta, c) V^T of src/draw-fstar.cpp:17 (exact to 1.3e-15
                               * for r >= 56) and solves r + m right-hand sides instead of 1001 + m; 0 = every grid column
                               * is solved */
    int      reserved[5];     /* must be 0.  (Up to version 100 kernel_fp32 and kstar_rank were the unnamed slots
                               * reserved[1] and reserved[2] of an int reserved[8] at this offset: same layout, same size.) */
} gpirt_options;

/* The reference's contract: GPIRT_RNG_RSTREAM (draw-for-draw replay of R's stream), draw_theta and draw_fstar as
 * written.  NOTE: these defaults REQUIRE an R stream -- gpirt_sampler_create / gpirt_mcmc called with opts == NULL
 * (or with unmodified defaults) and rs == NULL fail with GPIRT_E_ARG ("GPIRT_RNG_RSTREAM needs an R stream state");
 * a host without R's RNG state sets rng_kind = GPIRT_RNG_ITEM (and a seed) explicitly. */
void gpirt_default_options(gpirt_options* o);
/* The throughput preset -- what bench.py times as its headline and what BASELINE.json's metric is quoted on: the item-keyed
 * RNG (seed = 1: set your own), theta_stabilise = 1, fstar_fused = 1, kstar_rank = 64.  Same sampler, every form
 * algebraically identical to the reference's (DESIGN.md sections 4-5: f* within 1e-9 of the as-written form relative to
 * max|f*| at 8192 x 1024, checked in every bench.py run); draws are NOT those of R's stream (gpirt_default_options is
 * that contract).  One call away from the drop-in: the R shim selects it with options(gpirt.hip.preset = "fast"). */
void gpirt_fast_options(gpirt_options* o);

/* Whole-call drop-in for .gpirtMCMC (src/gpirtMCMC.cpp:5-117), all pointers HOST:
 *   h_y            n x m   responses (never modified)
 *   h_theta0       n       initial theta (copied, R semantics)
 *   h_prior_means / h_prior_sds / h_step_sizes   2 x m
 *   h_theta_draws  (S+1) x n, h_beta_draws 2 x m x (S+1), h_f_draws n x m x (S+1), h_irfs 1001 x m
 * R-stream mode: rs carries R's Mersenne-Twister state in and out (GetRNGstate/PutRNGstate of
 * Rcpp::RNGScope, src/RcppExports.cpp:19); ignored for GPIRT_RNG_ITEM.
 * tick may be NULL; it is called once per iteration before the draws (Rprintf +
 * checkUserInterrupt, src/gpirtMCMC.cpp:64-66,83-85). */
int gpirt_mcmc(const double* h_y, int64_t n, int64_t m, const double* h_theta0,
               int sample_iterations, int burn_iterations, const double* h_prior_means,
               const double* h_prior_sds, const double* h_step_sizes, const gpirt_options* opts,
               gpirt_rstream_t rs, gpirt_tick_fn tick, void* tick_ctx, double* h_theta_draws,
               double* h_beta_draws, double* h_f_draws, double* h_irfs);

/* Stage-level sampler for hosts that drive the loop themselves (bench.py, multi-GPU hosts that
 * put a collective between stages).  State lives on the device. */
int gpirt_sampler_create(gpirt_sampler_t* s, gpirt_handle_t h, const double* h_y, int64_t n,
                         int64_t m, const double* h_theta0, const double* h_prior_means,
                         const double* h_prior_sds, const double* h_step_sizes,
                         const gpirt_options* opts, gpirt_rstream_t rs);
int gpirt_sampler_destroy(gpirt_sampler_t s);
int gpirt_sampler_init(gpirt_sampler_t s);               /* src/gpirtMCMC.cpp:13-47 */
int gpirt_sampler_step(gpirt_sampler_t s);               /* one full iteration, :68-78 / :87-97 */
/* the stages of one iteration, in the reference's order */
int gpirt_sampler_draw_f(gpirt_sampler_t s);             /* :68 / :87 */
int gpirt_sampler_draw_fstar(gpirt_sampler_t s);         /* :69 / :88 */
int gpirt_sampler_theta_partial(gpirt_sampler_t s);      /* local-item part of draw_theta's log-posterior */
int gpirt_sampler_theta_finish(gpirt_sampler_t s);       /* :70 / :89 (after any cross-rank reduction) */
/* Respondent-block form of draw_theta for item-sharded runs (replaces theta_partial / all-reduce / theta_finish):
 * once, hand over this rank's block of the response matrix with ALL item columns (host, n_block x m_total,
 * column-major, same coding as y); per iteration gather every rank's f* columns into the device array
 * "fstar_full" (N x m_total, gpirt_sampler_devptr), call theta_block -- draw-theta.cpp:15-34 for the respondents
 * [i0, i0 + n_block), result in "theta_stage" (n values, zero outside the block) --, sum "theta_stage" over the
 * ranks and call theta_commit.  Draws are keyed by the global respondent index: independent of the partition. */
int gpirt_sampler_set_theta_block(gpirt_sampler_t s, const double* y_block, int64_t i0, int64_t n_block, int64_t m_total);
int gpirt_sampler_theta_block(gpirt_sampler_t s);
int gpirt_sampler_theta_commit(gpirt_sampler_t s);
int gpirt_sampler_draw_beta(gpirt_sampler_t s);          /* :71-75 / :90-94 (beta, mu, mu_star) */
int gpirt_sampler_factor(gpirt_sampler_t s);             /* :76-78 / :95-97 */
/* K(theta, theta) + jitter into "L" (lower blocks; src/gpirtMCMC.cpp:76-77) without factoring: the first step of a
 * factorisation the host distributes with the gpirt_potrf_panel_* pieces on the "L" devptr; gpirt_sampler_skip_factor
 * then closes the iteration. */
int gpirt_sampler_build_cov(gpirt_sampler_t s);
/* The pieces on this sampler's own "L" (which may carry extra rows below the n x n factor, see gpirt_sampler_ldl):
 * panel_rows = number of rows of a panel's column block that travel (n + extra - p W rows for panel p). */
int gpirt_sampler_panel_factor(gpirt_sampler_t s, int64_t p);
int gpirt_sampler_panel_update(gpirt_sampler_t s, int64_t p, int64_t c);
int gpirt_sampler_panel_copy(gpirt_sampler_t s, int64_t p, double* d_buf, int to_buf);
int gpirt_sampler_panel_rows(gpirt_sampler_t s, int64_t* rows);
/* ... and by halves of an outer panel (gpirt_potrf_panel_*_part above) */
int gpirt_sampler_panel_factor_part(gpirt_sampler_t s, int64_t p, int half);
int gpirt_sampler_panel_update_part(gpirt_sampler_t s, int64_t p, int64_t c, int part);
int gpirt_sampler_panel_copy_part(gpirt_sampler_t s, int64_t p, int half, double* d_buf, int64_t buf_doubles, int to_buf);
/* Close the iteration WITHOUT factoring: "L" arrived from elsewhere (a broadcast into the "L" devptr, the distributed
 * pieces, gpirt_sampler_set).  rows_with_L != 0: the rows below the n x n factor (gpirt_sampler_ldl) arrived with it,
 * i.e. the whole ldl x n buffer was received; 0: only the n x n factor is current and the rows are rebuilt by the
 * explicit forward solve (src/draw-fstar.cpp:19) before draw_fstar reads them.  gpirt_sampler_skip_factor(s) is
 * gpirt_sampler_adopt_factor(s, 0), the form that is safe for any host. */
int gpirt_sampler_adopt_factor(gpirt_sampler_t s, int rows_with_L);
int gpirt_sampler_skip_factor(gpirt_sampler_t s);
int gpirt_sampler_accumulate_irf(gpirt_sampler_t s);     /* :103 */
int gpirt_sampler_iteration(gpirt_sampler_t s, int* iter);
/* Sets the completed-iteration counter (the GPIRT_RNG_ITEM sub-streams are keyed by it): lets a second sampler
 * replay a stage of another one's iteration on copied state (bench.py's in-run check of the draw_fstar forms). */
int gpirt_sampler_set_iteration(gpirt_sampler_t s, int iter);
int gpirt_sampler_check(gpirt_sampler_t s);              /* syncs; returns potrf info / GPIRT_E_* */
/* Device pointer of a named state array ("theta","f","beta","mu","mu_star","fstar","L","logpost",
 * "irf_sum","ess_k") and its element count; the pointer stays valid until destroy. */
int gpirt_sampler_devptr(gpirt_sampler_t s, const char* name, void** d_ptr, int64_t* count);
/* Leading dimension of the "L" device array.  With n % 64 == 0 the array is (n + e) x n: e rows below the factor enter
 * the factorisation as K(c, theta) (rank-r K*, e = r) or K(theta*, theta) (e = 1024: the 1001 grid points, n >= 1024)
 * and leave it as (L^-1 K(theta, .))^T -- draw_fstar's forward solve (src/draw-fstar.cpp:19) comes out of a bordered
 * factorisation.
 * gpirt_sampler_get / _set("L") always move the n x n factor. */
int gpirt_sampler_ldl(gpirt_sampler_t s, int64_t* ldl);
/* dst's chain state := src's (theta, f, beta, mu, mu_star, fstar, L, iteration counter); same handle, same n and m. */
int gpirt_sampler_copy_state(gpirt_sampler_t dst, gpirt_sampler_t src);
int gpirt_sampler_get(gpirt_sampler_t s, const char* name, double* h_out, int64_t count);
int gpirt_sampler_set(gpirt_sampler_t s, const char* name, const double* h_in, int64_t count);
int gpirt_sampler_finish_irfs(gpirt_sampler_t s, int sample_iterations, double* h_irfs); /* :106-111 */
/* Per-stage device time of the last gpirt_sampler_step (ms, hipEvents); names_out is a
 * NUL-separated list terminated by an empty string. */
int gpirt_sampler_enable_timing(gpirt_sampler_t s, int on);
int gpirt_sampler_stage_times(gpirt_sampler_t s, double* ms_out, int max_stages, int* n_stages,
                              const char** names_out);
/* Device time and launch count of the potrf trailing-update kernel accumulated since the last
 * reset (hipEvents on the launch stream); the roofline figure of bench.py. */
int gpirt_prof_trailing(gpirt_handle_t h, int reset, double* total_ms, int64_t* launches,
                        double* flops);
/* The same per class of syrk launch inside arma::chol's replacement (src/gpirtMCMC.cpp:17,78,97):
 * cls 0 = trailing update on the 128-tile kernel (what gpirt_prof_trailing reports), 1 = trailing update on the
 * 64-tile kernel, 2 = the update between the two sub-panels of an outer panel (K = gpirt_potrf_subpanel_width(n)).  flops are the
 * algorithmic ones of the lower trapezoid, 2 K (M N - N (N - 1) / 2).
 * The same instrument on two kernels of draw_f: cls 3 = nu = L Z of the item-keyed draw_f (src/mvnormal.h:10 for all m
 * columns as one triangular product: n^2 m flops), 4 = the R-stream replay's pass over L (rs3_products_kernel: bytes = the
 * lower triangle of L, 8 n (n + 1) / 2, flops = 2 x 32 candidate columns x n (n + 1) / 2).
 * And on draw_theta: cls 5 = the log-posterior product in fixed point on the int8 matrix cores (csrc/theta_fixed.hip:
 * "flops" = the int8 multiply-adds x 2 of its seven digit planes, 7 x 2 x 1001 x n x 2m). */
int gpirt_prof_syrk(gpirt_handle_t h, int cls, int reset, double* total_ms, int64_t* launches, double* flops);
/* algorithmic bytes of the same launches (call before the resetting gpirt_prof_syrk): the C trapezoid read and
 * written once, the M x K panel operand read once -- what the launch must move if nothing is re-read */
int gpirt_prof_syrk_bytes(gpirt_handle_t h, int cls, double* bytes);
int gpirt_prof_enable(gpirt_handle_t h, int on);

#ifdef __cplusplus
}
#endif
#endif /* GPIRT_HIP_H */
