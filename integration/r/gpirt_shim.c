/*
 * gpirt_shim.c -- the plain-C `.Call` routine a maintainer of duckmayr/gpirt adds to swap the
 * RcppArmadillo sampler for libgpirt_hip.so.  It replaces src/RcppExports.cpp:16-40 (the generated
 * Rcpp glue `_gpirt_gpirtMCMC` + R_init_gpirt) one for one: same symbol, same 7 arguments, same
 * returned list (src/gpirtMCMC.cpp:112-116), same RNG bracketing (Rcpp::RNGScope,
 * src/RcppExports.cpp:19), same progress text and interrupt polling (src/gpirtMCMC.cpp:64-66,105).
 *
 * NOT built in the build container (R.h / Rinternals.h are absent there): tests/test_host.py only
 * checks that it still PARSES and type-checks, against declarations-only stand-ins for the R headers it names
 * (tests/r_stub/); every behaviour it relies on is exercised through the same C ABI by the Python harness
 * (gpirt_amd/sampler.py).  Build inside the R package with:
 *     PKG_CPPFLAGS = -I<repo>/include      PKG_LIBS = -L<repo>/gpirt_amd -lgpirt_hip
 */
#include <R.h>
#include <string.h>
#include <Rinternals.h>
#include <R_ext/Random.h>
#include <R_ext/Rdynload.h>
#include <R_ext/Utils.h>

#include "gpirt_hip.h"

static void chk_interrupt(void* dummy) { (void)dummy; R_CheckUserInterrupt(); }

/* Rprintf("\r%6.3f %% complete", progress); progress += progress_increment; Rcpp::checkUserInterrupt() --
 * src/gpirtMCMC.cpp:57-66, 83-85.  The reference ACCUMULATES progress_increment = (1.0 / total) * 100.0 once per
 * iteration, so the number printed in front of iteration `iter` is that increment added `iter` times (not
 * 100 * iter / total, which differs in the last printed digit now and then).  Recomputed from `iter` on every call:
 * the core may call tick again for an iteration it repeats after a hang-guard fallback (gpirt_mcmc). */
static int tick(void* ctx, int iter, int total)
{
    (void)ctx;
    const double progress_increment = (1.0 / (double)total) * 100.0;
    double progress = 0.0;
    for (int k = 0; k < iter; ++k) progress += progress_increment;
    Rprintf("\r%6.3f %% complete", progress);
    /* R_ToplevelExec returns FALSE when the user interrupted: ask the core to stop cleanly so
     * device memory is released before the R condition is raised */
    return R_ToplevelExec(chk_interrupt, NULL) ? 0 : 1;
}

/* R's Mersenne-Twister state lives in .Random.seed: [kind, mti, mt[0..623]].  The reference reads
 * and writes it through GetRNGstate()/PutRNGstate(); the HIP core replays the same stream, so the
 * state is handed over explicitly and written back. */
/* .Random.seed[1] = RNG kind + 100 * normal kind + 10000 * sample kind (R's RNG.c): Mersenne-Twister is kind 3 and
 * INVERSION is normal kind 4 -- BOTH are needed: under RNGkind(normal.kind = "Box-Muller") R::rnorm no longer is
 * qnorm of two uniforms, and replaying inversion normals would silently draw something else. */
static int rng_is_default_mt(SEXP seedvec)
{
    if (TYPEOF(seedvec) != INTSXP || LENGTH(seedvec) != 626) return 0;
    const int code = INTEGER(seedvec)[0];
    return (code % 100) == 3 && ((code / 100) % 100) == 4;
}

SEXP _gpirt_gpirtMCMC(SEXP ySEXP, SEXP thetaSEXP, SEXP sample_iterationsSEXP, SEXP burn_iterationsSEXP,
                      SEXP beta_prior_meansSEXP, SEXP beta_prior_sdsSEXP, SEXP beta_step_sizesSEXP)
{
    /* Rcpp's input_parameter<const arma::mat&> / <arma::vec> / <const int> (src/RcppExports.cpp:20-26) coerce
     * whatever storage mode R hands over (an integer-storage response_matrix passes is.response_matrix,
     * R/response_matrix.R:109-115) and as<arma::mat> insists on a matrix: do both by hand, PROTECTed, before
     * anything is dereferenced. */
    int nprot = 0;
    SEXP yR = PROTECT(coerceVector(ySEXP, REALSXP)); ++nprot;
    SEXP thR = PROTECT(coerceVector(thetaSEXP, REALSXP)); ++nprot;
    SEXP pmR = PROTECT(coerceVector(beta_prior_meansSEXP, REALSXP)); ++nprot;
    SEXP psR = PROTECT(coerceVector(beta_prior_sdsSEXP, REALSXP)); ++nprot;
    SEXP stR = PROTECT(coerceVector(beta_step_sizesSEXP, REALSXP)); ++nprot;
    SEXP dim = getAttrib(ySEXP, R_DimSymbol);
    if (TYPEOF(dim) != INTSXP || LENGTH(dim) != 2) { UNPROTECT(nprot); error("gpirt-hip: y must be a matrix"); }
    const int64_t n = INTEGER(dim)[0], m = INTEGER(dim)[1];
    if (n < 1 || m < 1) { UNPROTECT(nprot); error("gpirt-hip: y must have at least one row and one column"); }
    if ((int64_t)XLENGTH(thR) != n) { UNPROTECT(nprot); error("gpirt-hip: theta must have one value per row of y (%ld)", (long)n); }
    {
        SEXP mats[3] = { beta_prior_meansSEXP, beta_prior_sdsSEXP, beta_step_sizesSEXP };
        const char* nm[3] = { "beta_prior_means", "beta_prior_sds", "beta_step_sizes" };
        for (int k = 0; k < 3; ++k) {
            SEXP d = getAttrib(mats[k], R_DimSymbol);
            if (TYPEOF(d) != INTSXP || LENGTH(d) != 2 || INTEGER(d)[0] != 2 || (int64_t)INTEGER(d)[1] != m) {
                UNPROTECT(nprot);
                error("gpirt-hip: %s must be a 2 x %ld matrix", nm[k], (long)m);
            }
        }
    }
    const int S = asInteger(sample_iterationsSEXP), B = asInteger(burn_iterationsSEXP);
    if (S == NA_INTEGER || B == NA_INTEGER || S < 0 || B < 0) { UNPROTECT(nprot); error("gpirt-hip: iteration counts must be non-negative integers"); }
    const int64_t N = GPIRT_NGRID;

    SEXP theta = PROTECT(allocMatrix(REALSXP, S + 1, (int)n)); ++nprot;
    SEXP beta = PROTECT(alloc3DArray(REALSXP, 2, (int)m, S + 1)); ++nprot;
    SEXP f = PROTECT(alloc3DArray(REALSXP, (int)n, (int)m, S + 1)); ++nprot;
    SEXP irfs = PROTECT(allocMatrix(REALSXP, (int)N, (int)m)); ++nprot;

    gpirt_options opt;
    gpirt_default_options(&opt);                             /* = the reference's contract: R stream, no extras */
    /* options(gpirt.hip.preset = "fast"): the throughput preset of the library (gpirt_fast_options: item-keyed RNG,
     * stabilised draw_theta, fused + rank-64 draw_fstar -- what its benchmark is quoted on); single options below
     * still override it */
    SEXP preset = GetOption1(install("gpirt.hip.preset"));
    const int fast = isString(preset) && strcmp(CHAR(STRING_ELT(preset, 0)), "fast") == 0;
    if (fast) gpirt_fast_options(&opt);
    SEXP o1 = GetOption1(install("gpirt.hip.theta_stabilise"));
    if (o1 != R_NilValue) opt.theta_stabilise = asLogical(o1) == TRUE;
    SEXP o2 = GetOption1(install("gpirt.hip.fstar_fused"));
    if (o2 != R_NilValue) opt.fstar_fused = asLogical(o2) == TRUE;
    SEXP o3 = GetOption1(install("gpirt.hip.kstar_rank"));
    if (o3 != R_NilValue) opt.kstar_rank = asInteger(o3);
    if (opt.kstar_rank == NA_INTEGER || !opt.fstar_fused) opt.kstar_rank = 0;

    /* options(gpirt.hip.rng = "item") selects the batched counter-based contract; the default
     * replays R's own stream so results are draw-for-draw those of the RcppArmadillo build */
    SEXP rngopt = GetOption1(install("gpirt.hip.rng"));
    const int item_rng = isString(rngopt) ? strcmp(CHAR(STRING_ELT(rngopt, 0)), "item") == 0 : fast;

    gpirt_rstream_t rs = NULL;
    SEXP seedvec = R_NilValue;
    GetRNGstate();                                           /* Rcpp::RNGScope, entry */
    if (item_rng) {
        opt.rng_kind = GPIRT_RNG_ITEM;
        opt.seed = (uint64_t)(unif_rand() * 4294967296.0) << 32 | (uint64_t)(unif_rand() * 4294967296.0);
    } else {
        opt.rng_kind = GPIRT_RNG_RSTREAM;
        PutRNGstate();                                       /* make .Random.seed current */
        seedvec = findVarInFrame(R_GlobalEnv, install(".Random.seed"));
        if (!rng_is_default_mt(seedvec)) {
            UNPROTECT(nprot);
            error("gpirt-hip: rng = \"reference\" needs RNGkind(\"Mersenne-Twister\", \"Inversion\")");
        }
        gpirt_rstream_from_state(&rs, (const uint32_t*)(INTEGER(seedvec) + 2), INTEGER(seedvec)[1]);
        GetRNGstate();
    }

    int rc = gpirt_mcmc(REAL(yR), n, m, REAL(thR), S, B, REAL(pmR), REAL(psR), REAL(stR), &opt, rs, tick, NULL,
                        REAL(theta), REAL(beta), REAL(f), REAL(irfs));

    if (rs) {                                                /* hand the advanced stream back to R */
        int mti = 0;
        gpirt_rstream_get_state(rs, (uint32_t*)(INTEGER(seedvec) + 2), &mti);
        INTEGER(seedvec)[1] = mti;
        gpirt_rstream_destroy(rs);
        GetRNGstate();                                       /* re-read the modified .Random.seed */
    }
    PutRNGstate();                                           /* Rcpp::RNGScope, exit */

    if (rc != 0) {                                           /* device resources are already released */
        UNPROTECT(nprot);
        if (rc == GPIRT_E_INTERRUPT) { Rprintf("\n"); Rf_onintr(); }
        error("%s", rc > 0 ? "chol(): decomposition failed" : gpirt_last_error());
    }
    Rprintf("\r100.000 %% complete\n");                      /* src/gpirtMCMC.cpp:105 */

    SEXP res = PROTECT(allocVector(VECSXP, 4)); ++nprot;
    SEXP names = PROTECT(allocVector(STRSXP, 4)); ++nprot;
    SET_VECTOR_ELT(res, 0, theta); SET_STRING_ELT(names, 0, mkChar("theta"));
    SET_VECTOR_ELT(res, 1, beta);  SET_STRING_ELT(names, 1, mkChar("beta"));
    SET_VECTOR_ELT(res, 2, f);     SET_STRING_ELT(names, 2, mkChar("f"));
    SET_VECTOR_ELT(res, 3, irfs);  SET_STRING_ELT(names, 3, mkChar("IRFs"));
    setAttrib(res, R_NamesSymbol, names);
    UNPROTECT(nprot);
    return res;
}

static const R_CallMethodDef CallEntries[] = {
    {"_gpirt_gpirtMCMC", (DL_FUNC)&_gpirt_gpirtMCMC, 7},      /* src/RcppExports.cpp:32-35 */
    {NULL, NULL, 0}
};

void R_init_gpirt(DllInfo* dll)                               /* src/RcppExports.cpp:37-40 */
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
