/*
 * gpirt_oracle.h -- CPU restatement of duckmayr/gpirt's sampler, used ONLY as the parity
 * checker (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Nothing under gpirt_amd/ may include, link,
 * import or call it.  The product path is the HIP library (include/gpirt_hip.h).
 *
 * Parity status: the reference cannot be built or run here (no R, Rcpp, Armadillo, Rmath, BLAS or
 * LAPACK in the image) and its own tests pin no sampler output, so the
 * oracle is a line-following restatement pinned by
 *   (i)  known-answer tests of R's RNG (set.seed / runif / rnorm / qnorm public values), and
 *   (ii) an independent NumPy/SciPy(LAPACK) statement of the same stages (oracle/np_oracle.py).
 * i.e. "parity unpinned" by reference golden vectors; see DESIGN.md.
 *
 * All matrices are column-major doubles (Armadillo / R layout).  y holds -1 / +1 / NaN.
 * Every function cites the reference file:line (paths relative to the upstream repo root).
 */
#ifndef GPIRT_ORACLE_H
#define GPIRT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ RNG ---------------- */

enum { ORC_RNG_RSTREAM = 0, ORC_RNG_ITEM = 1 };

/* stage ids of the counter-based ("item") RNG contract; shared verbatim with the HIP side */
enum {
    ORC_ST_INIT_F    = 1,  /* initial f_j = L z_j          index = respondent i          */
    ORC_ST_INIT_BETA = 2,  /* initial beta(p,j)            index = p                     */
    ORC_ST_F_Z       = 3,  /* ess(): nu = L z              index = respondent i          */
    ORC_ST_F_ESS     = 4,  /* ess(): u, eps0, eps_r...     index = 0,1,2+r               */
    ORC_ST_FSTAR     = 5,  /* draw_fstar normals           index = grid point i          */
    ORC_ST_THETA     = 6,  /* draw_theta uniform           item = respondent, index = 0  */
    ORC_ST_BETA      = 7   /* draw_beta: normal 2k, unif 2k+1                            */
};

typedef struct orc_rng {
    int      kind;
    /* R stream: Mersenne-Twister state exactly as R keeps it (dummy[0]=mti, dummy[1..624]=mt) */
    uint32_t mt[624];
    int      mti;
    uint64_t n_unif;      /* uniforms consumed so far (diagnostic; pins control flow)        */
    /* item stream: Philox4x32-10, key = seed, counter = (index, item, stage, iteration)     */
    uint64_t seed;
    uint32_t iter;
    uint32_t stage;
    uint32_t item;
    uint32_t index;       /* next index inside the current sub-stream                        */
    uint32_t item_base;   /* global index of local item 0 (item-sharded runs; 0 otherwise)   */
    uint32_t item_local;  /* local item whose ess() is running                               */
} orc_rng;

void   orc_rng_init_rstream(orc_rng* r, uint32_t seed);          /* == set.seed(seed)        */
void   orc_rng_init_rstate(orc_rng* r, const uint32_t mt[624], int mti);
void   orc_rng_init_item(orc_rng* r, uint64_t seed);
void   orc_rng_substream(orc_rng* r, uint32_t iter, uint32_t stage, uint32_t item);
double orc_unif_rand(orc_rng* r);
double orc_norm_rand(orc_rng* r);
double orc_rnorm(orc_rng* r, double mu, double sd);              /* R::rnorm                 */
double orc_runif(orc_rng* r, double a, double b);                /* R::runif                 */
double orc_qnorm(double p);                                      /* R::qnorm(p,0,1,1,0)      */
double orc_dnorm_log(double x, double mu, double sd);            /* R::dnorm(x,mu,sd,1)      */
double orc_plogis(double x);                                     /* R::plogis(x,0,1,1,0)     */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_item_uniform(uint64_t seed, uint32_t iter, uint32_t stage, uint32_t item, uint32_t index);

/* ------------------------------------------------------------------ stages ------------- */

/* K(): src/covariance-function.cpp:3-14.  out is n1 x n2, leading dimension n1.             */
void orc_se_kernel(const double* x1, int64_t n1, const double* x2, int64_t n2, double* out);

/* S.diag() += jitter; cholS = arma::chol(S,"lower"): src/gpirtMCMC.cpp:15-17.
 * In place on the n x n matrix A (full symmetric on entry, lower factor + zeroed upper on exit).
 * Returns 0, or j+1 if the leading minor of order j+1 is not positive definite (LAPACK info).  */
int  orc_potrf_lower(double* A, int64_t n);            /* unblocked, deterministic            */
int  orc_potrf_lower_blocked(double* A, int64_t n, int nthreads); /* cpu_baseline speed      */

/* rmvnorm(): src/mvnormal.h:4-11.  z (n) is filled from the rng, out = L z (dense gemv).     */
void orc_rmvnorm(orc_rng* r, const double* L, int64_t n, double* z, double* out);

/* ll(), ll_bar(): src/log-likelihood.cpp:12-23, 25-37 */
double orc_ll(const double* f, const double* y, int64_t n);
double orc_ll_bar(const double* f, const double* y, const double* mu, int64_t n);

/* ess(): src/draw-f.cpp:21-60.  Returns the number of rejections k. Optional trace outputs. */
typedef struct orc_ess_trace { double u, log_y, eps0, eps_final; int k; } orc_ess_trace;
int  orc_ess(orc_rng* r, const double* f, const double* y, const double* L, const double* mu,
             int64_t n, double* f_out, double* nu_out, orc_ess_trace* tr);

/* draw_f(): src/draw-f.cpp:64-73.  iter is only used to key the item-stream RNG.             */
void orc_draw_f(orc_rng* r, uint32_t iter, const double* f, const double* y, const double* L,
                const double* mu, int64_t n, int64_t m, double* f_out, int* k_out);

/* solve(trimatl(L), B) / solve(trimatu(L.t()), B): src/draw-fstar.cpp:7,19. In place on B.   */
void orc_trsm_lower(const double* L, int64_t n, double* B, int64_t nrhs, int trans);

/* draw_fstar(): src/draw-fstar.cpp:10-31.  out N x m.  Optional s_out (N), mean_out (N x m). */
void orc_draw_fstar(orc_rng* r, uint32_t iter, const double* f, const double* theta,
                    const double* theta_star, const double* L, const double* mu_star,
                    int64_t n, int64_t m, int64_t N, double* out, double* s_out, double* mean_out);

/* The two halves of draw_fstar() separately (the all-core parity driver spreads grid columns and items over threads):
 * :17-20 for any slice of the grid (kstar_out n x N required, tmp_out optional), and :23-29 for any slice of items.   */
void orc_fstar_grid(const double* theta, const double* theta_star, const double* L, int64_t n, int64_t N,
                    double* kstar_out, double* tmp_out, double* s_out);
void orc_fstar_items(orc_rng* r, uint32_t iter, const double* f, const double* kstar, const double* s,
                     const double* L, const double* mu_star, int64_t n, int64_t m, int64_t N,
                     double* out, double* mean_out);

/* draw_theta(): src/draw-theta.cpp:3-37.  stabilise!=0 subtracts the row maximum before exp
 * (identical up to rounding wherever the reference is defined).  Returns the
 * number of respondents whose CDF degenerated (reference would read theta_star[N], UB).      */
int  orc_draw_theta(orc_rng* r, uint32_t iter, const double* theta_star, const double* y,
                    const double* theta_prior, const double* fstar, int64_t n, int64_t m,
                    int64_t N, int stabilise, double* theta_out);

/* ... for the block of respondents i0 .. i0 + n - 1 (y n x m holds just those rows): same draws as the whole loop */
int  orc_draw_theta_block(orc_rng* r, uint32_t iter, const double* theta_star, const double* y,
                          const double* theta_prior, const double* fstar, int64_t n, int64_t m,
                          int64_t N, int stabilise, int64_t i0, double* theta_out);

/* draw_beta(): src/draw-beta.cpp:3-41.  X = [1, theta].                                      */
void orc_draw_beta(orc_rng* r, uint32_t iter, const double* beta, const double* theta,
                   const double* y, const double* f, const double* prior_means,
                   const double* prior_sds, const double* step_sizes, int64_t n, int64_t m,
                   double* beta_out);

/* gpirtMCMC(): src/gpirtMCMC.cpp:5-117.
 * Outputs (column-major): theta_draws (S+1) x n, beta_draws 2 x m x (S+1), f_draws n x m x (S+1),
 * IRFs N x m (N = 1001).  Returns 0 or potrf info (>0).                                      */
typedef struct orc_mcmc_opts {
    int theta_stabilise;      /* 0 = verbatim reference arithmetic, 1 = row-max shift          */
    int blocked_potrf;        /* 0 = unblocked (deterministic restatement), 1 = blocked        */
    int nthreads;             /* for blocked potrf / timing legs                               */
    int fstar_fused;          /* 0 = double_solve as in the reference; 1 = tmp^T (L^-1 f) form */
} orc_mcmc_opts;

int orc_gpirt_mcmc(orc_rng* r, const double* y, int64_t n, int64_t m, const double* theta0,
                   int sample_iterations, int burn_iterations, const double* beta_prior_means,
                   const double* beta_prior_sds, const double* beta_step_sizes,
                   const orc_mcmc_opts* opts, double* theta_draws, double* beta_draws,
                   double* f_draws, double* IRFs,
                   /* optional final-state outputs for stage-level parity (may be NULL) */
                   double* L_final, double* fstar_final);

#define ORC_NGRID 1001
void orc_theta_star_grid(double* theta_star /* 1001 */);       /* src/gpirtMCMC.cpp:35       */

#ifdef __cplusplus
}
#endif
#endif
