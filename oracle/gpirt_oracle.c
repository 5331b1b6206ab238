/*
 * gpirt_oracle.c -- CPU restatement of the duckmayr/gpirt sampler (TEST INFRASTRUCTURE ONLY).
 *
 * See gpirt_oracle.h for the contract and the parity status ("parity unpinned" by reference
 * golden vectors: the reference ships none and cannot be run here; pinned by R-RNG known-answer
 * tests and an independent NumPy/SciPy statement).
 *
 * Compile with -ffp-contract=off so no FMA contraction changes the arithmetic between hosts.
 * Reference paths cited below are relative to the upstream repository root.
 */
#include "gpirt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_2PI 6.283185307179586476925286766559 /* Rmath.h M_2PI, used at src/draw-f.cpp:34,36 */
#define ORC_LN_SQRT_2PI 0.918938533204672741780329736406

/* ======================================================================================
 * R's default RNG: Mersenne-Twister + inversion.  Third-party arithmetic NOT vendored under
 * /root/reference (R >= 3.4.0, DESCRIPTION:25-26); restated from R's published algorithm
 * (src/main/RNG.c: MT_sgenrand/MT_genrand/fixup/Randomize; src/nmath/snorm.c INVERSION;
 * src/nmath/qnorm.c = Wichura AS241 PPND16).  Pinned by tests/test_oracle_rng.py KATs.
 * Call sites in the reference: mvnormal.h:8, draw-f.cpp:28,35,56, draw-fstar.cpp:27,
 * draw-theta.cpp:27, draw-beta.cpp:22,30, gpirtMCMC.cpp:25.
 * ====================================================================================== */

void orc_rng_init_rstream(orc_rng* r, uint32_t seed)
{
    memset(r, 0, sizeof(*r));
    r->kind = ORC_RNG_RSTREAM;
    /* Randomize(): 50 scrambling rounds, then 625 LCG outputs fill dummy[0..624];
     * FixupSeeds sets dummy[0] = 624 so the first draw regenerates the block. */
    for (int j = 0; j < 50; ++j) seed = 69069u * seed + 1u;
    uint32_t dummy[625];
    for (int j = 0; j < 625; ++j) { seed = 69069u * seed + 1u; dummy[j] = seed; }
    memcpy(r->mt, dummy + 1, sizeof(r->mt));
    r->mti = 624;
}

void orc_rng_init_rstate(orc_rng* r, const uint32_t mt[624], int mti)
{
    memset(r, 0, sizeof(*r));
    r->kind = ORC_RNG_RSTREAM;
    memcpy(r->mt, mt, sizeof(r->mt));
    r->mti = mti;
}

static uint32_t mt_next(orc_rng* r)
{
    enum { N = 624, M = 397 };
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    uint32_t* mt = r->mt;
    if (r->mti >= N) {
        int kk;
        uint32_t y;
        for (kk = 0; kk < N - M; ++kk) {
            y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        for (; kk < N - 1; ++kk) {
            y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        r->mti = 0;
    }
    uint32_t y = mt[r->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* Philox4x32-10 (Salmon et al. 2011), the counter-based generator of the "item" RNG contract. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
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

/* One uniform in (0,1) per (seed, iteration, stage, item, index): 52 random bits + 1/2 ulp. */
double orc_item_uniform(uint64_t seed, uint32_t iter, uint32_t stage, uint32_t item, uint32_t index)
{
    uint32_t ctr[4] = { index, item, stage, iter };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    uint64_t v = ((uint64_t)(o[0] >> 6) << 26) | (uint64_t)(o[1] >> 6);
    return ((double)v + 0.5) * 2.220446049250313e-16; /* 2^-52 */
}

void orc_rng_init_item(orc_rng* r, uint64_t seed)
{
    memset(r, 0, sizeof(*r));
    r->kind = ORC_RNG_ITEM;
    r->seed = seed;
}

void orc_rng_substream(orc_rng* r, uint32_t iter, uint32_t stage, uint32_t item)
{
    if (r->kind != ORC_RNG_ITEM) return; /* the R stream is one global sequence */
    /* item-keyed stages use the GLOBAL item index; draw_theta is keyed by respondent */
    r->iter = iter; r->stage = stage; r->index = 0;
    r->item = item + (stage != ORC_ST_THETA ? r->item_base : 0u);
}

double orc_unif_rand(orc_rng* r)
{
    r->n_unif++;
    if (r->kind == ORC_RNG_ITEM)
        return orc_item_uniform(r->seed, r->iter, r->stage, r->item, r->index++);
    const double i2_32m1 = 2.328306437080797e-10; /* 1/(2^32 - 1) */
    double v = (double)mt_next(r) * 2.3283064365386963e-10; /* [0,1) */
    if (v <= 0.0) return 0.5 * i2_32m1;
    if ((1.0 - v) <= 0.0) return 1.0 - 0.5 * i2_32m1;
    return v;
}

double orc_qnorm(double p)
{
    if (isnan(p)) return p;
    if (p <= 0.0) return p == 0.0 ? -INFINITY : NAN;
    if (p >= 1.0) return p == 1.0 ? INFINITY : NAN;
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

double orc_norm_rand(orc_rng* r)
{
    if (r->kind == ORC_RNG_ITEM) return orc_qnorm(orc_unif_rand(r));
    const double BIG = 134217728.0; /* 2^27 */
    double u = orc_unif_rand(r);
    u = (double)(int)(BIG * u) + orc_unif_rand(r);
    return orc_qnorm(u / BIG);
}

double orc_rnorm(orc_rng* r, double mu, double sd)
{
    if (isnan(mu) || !isfinite(sd) || sd < 0.0) return NAN;
    if (sd == 0.0 || !isfinite(mu)) return mu;   /* no RNG consumption */
    return mu + sd * orc_norm_rand(r);
}

double orc_runif(orc_rng* r, double a, double b)
{
    if (!isfinite(a) || !isfinite(b) || b < a) return NAN;
    if (a == b) return a;
    double u = orc_unif_rand(r);
    return a + (b - a) * u;
}

double orc_dnorm_log(double x, double mu, double sd)
{
    double z = (x - mu) / sd;
    z = fabs(z);
    return -(ORC_LN_SQRT_2PI + 0.5 * z * z + log(sd));
}

double orc_plogis(double x) { return 1.0 / (1.0 + exp(-x)); }

/* ====================================================================================== */

void orc_theta_star_grid(double* theta_star)
{
    /* arma::regspace<vec>(-5.0, 0.01, 5.0): src/gpirtMCMC.cpp:35 -> 1001 points start + i*delta */
    for (int i = 0; i < ORC_NGRID; ++i) theta_star[i] = -5.0 + (double)i * 0.01;
}

/* src/covariance-function.cpp:3-14 */
void orc_se_kernel(const double* x1, int64_t n1, const double* x2, int64_t n2, double* out)
{
    for (int64_t j = 0; j < n2; ++j)
        for (int64_t i = 0; i < n1; ++i) {
            double diff = x1[i] - x2[j];
            out[i + j * n1] = exp(-0.5 * diff * diff);
        }
}

/* arma::chol(S,"lower") -> LAPACK dpotrf('L'): src/gpirtMCMC.cpp:17,78,97.
 * Unblocked left-looking column Cholesky (dpotf2 ordering); upper triangle zeroed like Armadillo. */
int orc_potrf_lower(double* A, int64_t n)
{
    for (int64_t j = 0; j < n; ++j) {
        double* cj = A + j * n;
        /* column j -= sum_k L(j,k) * L(:,k) */
        for (int64_t k = 0; k < j; ++k) {
            const double* ck = A + k * n;
            double ljk = ck[j];
            if (ljk == 0.0) continue;
            for (int64_t i = j; i < n; ++i) cj[i] -= ljk * ck[i];
        }
        double d = cj[j];
        if (!(d > 0.0)) return (int)(j + 1);
        d = sqrt(d);
        cj[j] = d;
        for (int64_t i = j + 1; i < n; ++i) cj[i] /= d;
    }
    for (int64_t j = 1; j < n; ++j)
        for (int64_t i = 0; i < j; ++i) A[i + j * n] = 0.0;
    return 0;
}

/* src/mvnormal.h:4-11 : z_i = R::rnorm(0,1) in order, return cholS * z (dense product). */
void orc_rmvnorm(orc_rng* r, const double* L, int64_t n, double* z, double* out)
{
    for (int64_t i = 0; i < n; ++i) z[i] = orc_rnorm(r, 0.0, 1.0);
    for (int64_t i = 0; i < n; ++i) out[i] = 0.0;
    for (int64_t k = 0; k < n; ++k) {          /* dgemv 'N': axpy over columns */
        double zk = z[k];
        const double* col = L + k * n;
        for (int64_t i = k; i < n; ++i) out[i] += col[i] * zk; /* rows < k are exact zeros */
    }
}

/* src/log-likelihood.cpp:12-23 */
double orc_ll(const double* f, const double* y, int64_t n)
{
    double result = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (isnan(y[i])) continue;
        double a = y[i] * f[i];
        result -= log(1 + exp(-a));
    }
    return result;
}

/* src/log-likelihood.cpp:25-37 */
double orc_ll_bar(const double* f, const double* y, const double* mu, int64_t n)
{
    double result = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (isnan(y[i])) continue;
        double g = f[i] + mu[i];
        double a = y[i] * g;
        result -= log(1 + exp(-a));
    }
    return result;
}

/* src/draw-f.cpp:21-60 (quirk Q1: only epsilon_min is reset after the first draw) */
int orc_ess(orc_rng* r, const double* f, const double* y, const double* L, const double* mu,
            int64_t n, double* f_out, double* nu_out, orc_ess_trace* tr)
{
    double* z  = (double*)malloc(sizeof(double) * (size_t)n);
    double* nu = nu_out ? nu_out : (double*)malloc(sizeof(double) * (size_t)n);
    uint32_t iter = r->iter, item = r->item_local;
    orc_rng_substream(r, iter, ORC_ST_F_Z, item);
    orc_rmvnorm(r, L, n, z, nu);                                   /* :26 */
    orc_rng_substream(r, iter, ORC_ST_F_ESS, item);
    double u = orc_runif(r, 0.0, 1.0);                             /* :28 */
    double log_y = orc_ll_bar(f, y, mu, n) + log(u);               /* :29 */
    double eps_min = 0.0, eps_max = ORC_2PI;                       /* :33-34 */
    double eps = orc_runif(r, eps_min, eps_max);                   /* :35 */
    double eps0 = eps;
    eps_min = eps - ORC_2PI;                                       /* :36 */
    int k = 0;
    for (;;) {
        double c = cos(eps), s = sin(eps);
        for (int64_t i = 0; i < n; ++i) f_out[i] = f[i] * c + nu[i] * s;   /* :43 */
        if (orc_ll_bar(f_out, y, mu, n) > log_y) break;            /* :45-47 */
        if (eps < 0.0) eps_min = eps; else eps_max = eps;          /* :50-55 */
        eps = orc_runif(r, eps_min, eps_max);                      /* :56 */
        ++k;
    }
    if (tr) { tr->u = u; tr->log_y = log_y; tr->eps0 = eps0; tr->eps_final = eps; tr->k = k; }
    free(z);
    if (!nu_out) free(nu);
    return k;
}

/* src/draw-f.cpp:64-73 */
void orc_draw_f(orc_rng* r, uint32_t iter, const double* f, const double* y, const double* L,
                const double* mu, int64_t n, int64_t m, double* f_out, int* k_out)
{
    for (int64_t j = 0; j < m; ++j) {
        r->iter = iter; r->item_local = (uint32_t)j;
        int k = orc_ess(r, f + j * n, y + j * n, L, mu + j * n, n, f_out + j * n, NULL, NULL);
        if (k_out) k_out[j] = k;
    }
}

/* solve(trimatl(L), B) (trans=0) and solve(trimatu(L.t()), B) (trans=1): draw-fstar.cpp:7,19
 * -> LAPACK dtrtrs -> dtrsm; column-oriented substitution per right-hand side. */
void orc_trsm_lower(const double* L, int64_t n, double* B, int64_t nrhs, int trans)
{
    for (int64_t c = 0; c < nrhs; ++c) {
        double* b = B + c * n;
        if (!trans) {
            for (int64_t k = 0; k < n; ++k) {
                if (b[k] != 0.0) {
                    b[k] /= L[k + k * n];
                    double bk = b[k];
                    const double* col = L + k * n;
                    for (int64_t i = k + 1; i < n; ++i) b[i] -= bk * col[i];
                }
            }
        } else {
            for (int64_t k = n - 1; k >= 0; --k) {
                const double* col = L + k * n;
                double t = b[k];
                for (int64_t i = k + 1; i < n; ++i) t -= col[i] * b[i];
                b[k] = t / col[k];
            }
        }
    }
}

/* src/draw-fstar.cpp:17-20, the part that does not depend on the item: kstar = K(theta, theta_star) (n x N),
 * tmp = solve(trimatl(L), kstar), s = 1 - sqrt(colsum(tmp^2)) (quirk Q2).  theta_star may be any slice of the grid
 * (the columns are independent), which is how the all-core parity driver (tests/_oracle_parallel.py) spreads it. */
void orc_fstar_grid(const double* theta, const double* theta_star, const double* L, int64_t n, int64_t N,
                    double* kstar_out, double* tmp_out, double* s_out)
{
    double* tmp = tmp_out ? tmp_out : (double*)malloc(sizeof(double) * (size_t)(n * N));
    orc_se_kernel(theta, n, theta_star, N, kstar_out);             /* :17 */
    memcpy(tmp, kstar_out, sizeof(double) * (size_t)(n * N));
    orc_trsm_lower(L, n, tmp, N, 0);                               /* :19 */
    for (int64_t i = 0; i < N; ++i) {                              /* :20 */
        double q = 0.0;
        const double* t = tmp + i * n;
        for (int64_t k = 0; k < n; ++k) q += t[k] * t[k];
        s_out[i] = 1.0 - sqrt(q);
    }
    if (!tmp_out) free(tmp);
}

/* src/draw-fstar.cpp:23-29, the loop over items, given kstar (n x N) and s (N) from orc_fstar_grid.
 * Items are keyed (iter, ORC_ST_FSTAR, item_base + j): a slice of the items gives the same draws as the whole loop. */
void orc_fstar_items(orc_rng* r, uint32_t iter, const double* f, const double* kstar, const double* s,
                     const double* L, const double* mu_star, int64_t n, int64_t m, int64_t N,
                     double* out, double* mean_out)
{
    double* alpha = (double*)malloc(sizeof(double) * (size_t)n);
    for (int64_t j = 0; j < m; ++j) {                              /* :23-29 */
        memcpy(alpha, f + j * n, sizeof(double) * (size_t)n);
        orc_trsm_lower(L, n, alpha, 1, 0);                         /* :24 -> :7 */
        orc_trsm_lower(L, n, alpha, 1, 1);
        orc_rng_substream(r, iter, ORC_ST_FSTAR, (uint32_t)j);
        for (int64_t i = 0; i < N; ++i) {
            const double* kc = kstar + i * n;                      /* row i of kstarT */
            double acc = 0.0;
            for (int64_t k = 0; k < n; ++k) acc += kc[k] * alpha[k];
            double mean = acc + mu_star[i + j * N];                /* :25 */
            if (mean_out) mean_out[i + j * N] = mean;
            if (r->kind == ORC_RNG_ITEM) r->index = (uint32_t)i;   /* index-addressed */
            out[i + j * N] = orc_rnorm(r, mean, s[i]);             /* :27 */
        }
    }
    free(alpha);
}

/* src/draw-fstar.cpp:10-31 (quirk Q2: s = 1 - sqrt(q)) */
void orc_draw_fstar(orc_rng* r, uint32_t iter, const double* f, const double* theta,
                    const double* theta_star, const double* L, const double* mu_star,
                    int64_t n, int64_t m, int64_t N, double* out, double* s_out, double* mean_out)
{
    double* kstar = (double*)malloc(sizeof(double) * (size_t)(n * N));
    double* s     = (double*)malloc(sizeof(double) * (size_t)N);
    orc_fstar_grid(theta, theta_star, L, n, N, kstar, NULL, s);    /* :17-20 */
    if (s_out) memcpy(s_out, s, sizeof(double) * (size_t)N);
    orc_fstar_items(r, iter, f, kstar, s, L, mu_star, n, m, N, out, mean_out);   /* :23-29 */
    free(kstar); free(s);
}

/* Same stage with the algebraically identical form mean = (L^-1 kstar)^T (L^-1 f) + mu_star,
 * which drops the L^-T solve.  Used to price the difference (tests), not part of the reference. */
static void orc_draw_fstar_fused(orc_rng* r, uint32_t iter, const double* f, const double* theta,
                                 const double* theta_star, const double* L, const double* mu_star,
                                 int64_t n, int64_t m, int64_t N, double* out)
{
    double* tmp = (double*)malloc(sizeof(double) * (size_t)(n * N));
    double* s   = (double*)malloc(sizeof(double) * (size_t)N);
    double* w   = (double*)malloc(sizeof(double) * (size_t)n);
    orc_se_kernel(theta, n, theta_star, N, tmp);
    orc_trsm_lower(L, n, tmp, N, 0);
    for (int64_t i = 0; i < N; ++i) {
        double q = 0.0;
        const double* t = tmp + i * n;
        for (int64_t k = 0; k < n; ++k) q += t[k] * t[k];
        s[i] = 1.0 - sqrt(q);
    }
    for (int64_t j = 0; j < m; ++j) {
        memcpy(w, f + j * n, sizeof(double) * (size_t)n);
        orc_trsm_lower(L, n, w, 1, 0);
        orc_rng_substream(r, iter, ORC_ST_FSTAR, (uint32_t)j);
        for (int64_t i = 0; i < N; ++i) {
            const double* t = tmp + i * n;
            double acc = 0.0;
            for (int64_t k = 0; k < n; ++k) acc += t[k] * w[k];
            double mean = acc + mu_star[i + j * N];
            if (r->kind == ORC_RNG_ITEM) r->index = (uint32_t)i;
            out[i + j * N] = orc_rnorm(r, mean, s[i]);
        }
    }
    free(tmp); free(s); free(w);
}

/* src/draw-theta.cpp:3-37 (quirk Q5: P[0] rescales to 0; theta_star[N] is out of bounds) */
int orc_draw_theta(orc_rng* r, uint32_t iter, const double* theta_star, const double* y,
                   const double* theta_prior, const double* fstar, int64_t n, int64_t m,
                   int64_t N, int stabilise, double* theta_out)
{
    return orc_draw_theta_block(r, iter, theta_star, y, theta_prior, fstar, n, m, N, stabilise, 0, theta_out);
}

/* the same loop for a block of respondents: y holds rows i0 .. i0 + n - 1 (n x m, leading dimension n), the uniform of
 * local respondent i is keyed by its global index i0 + i (item RNG), so blocks reproduce the whole loop's draws */
int orc_draw_theta_block(orc_rng* r, uint32_t iter, const double* theta_star, const double* y,
                         const double* theta_prior, const double* fstar, int64_t n, int64_t m,
                         int64_t N, int stabilise, int64_t i0, double* theta_out)
{
    double* P = (double*)malloc(sizeof(double) * (size_t)N);
    int degenerate = 0;
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t k = 0; k < N; ++k) {
            /* ll(fstar.row(k).t(), y.row(i).t()) : :18 -> log-likelihood.cpp:12-23 */
            double res = 0.0;
            for (int64_t j = 0; j < m; ++j) {
                double yy = y[i + j * n];
                if (isnan(yy)) continue;
                double a = yy * fstar[k + j * N];
                res -= log(1 + exp(-a));
            }
            P[k] = theta_prior[k] + res;
        }
        if (stabilise) {
            double mx = -INFINITY;
            for (int64_t k = 0; k < N; ++k) if (P[k] > mx) mx = P[k];
            for (int64_t k = 0; k < N; ++k) P[k] -= mx;
        }
        for (int64_t k = 0; k < N; ++k) P[k] = exp(P[k]);            /* :21 */
        for (int64_t k = 1; k < N; ++k) P[k] = P[k - 1] + P[k];      /* :22 cumsum */
        double max_p = P[0], min_p = P[0];                           /* :23-24 */
        for (int64_t k = 1; k < N; ++k) { if (P[k] > max_p) max_p = P[k]; if (P[k] < min_p) min_p = P[k]; }
        for (int64_t k = 0; k < N; ++k) P[k] = (P[k] - min_p) / (max_p - min_p); /* :25 */
        orc_rng_substream(r, iter, ORC_ST_THETA, (uint32_t)(i0 + i));
        double u = orc_runif(r, 0.0, 1.0);                           /* :27 */
        double res = NAN;   /* reference: theta_star[N] (out of bounds, UB); we return NaN */
        int found = 0;
        for (int64_t k = 0; k < N; ++k) if (P[k] > u) { res = theta_star[k]; found = 1; break; }
        if (!found) ++degenerate;
        theta_out[i] = res;
    }
    free(P);
    return degenerate;
}

/* src/draw-beta.cpp:3-41 */
void orc_draw_beta(orc_rng* r, uint32_t iter, const double* beta, const double* theta,
                   const double* y, const double* f, const double* prior_means,
                   const double* prior_sds, const double* step_sizes, int64_t n, int64_t m,
                   double* beta_out)
{
    double* mu_p = (double*)malloc(sizeof(double) * (size_t)n);
    double* mu_c = (double*)malloc(sizeof(double) * (size_t)n);
    for (int64_t j = 0; j < m; ++j) {
        double cv[2] = { beta[0 + 2 * j], beta[1 + 2 * j] };
        double pv[2] = { cv[0], cv[1] };
        orc_rng_substream(r, iter, ORC_ST_BETA, (uint32_t)j);
        for (int k = 0; k < 2; ++k) {
            if (r->kind == ORC_RNG_ITEM) r->index = (uint32_t)(2 * k);
            pv[k] = orc_rnorm(r, cv[k], step_sizes[k + 2 * j]);                /* :22 */
            double pm = prior_means[k + 2 * j], ps = prior_sds[k + 2 * j];
            double pv_prior = orc_dnorm_log(pv[k], pm, ps);                    /* :25 */
            double cv_prior = orc_dnorm_log(cv[k], pm, ps);                    /* :26 */
            for (int64_t i = 0; i < n; ++i) {                                  /* X * pv, X * cv */
                mu_p[i] = pv[0] + theta[i] * pv[1];
                mu_c[i] = cv[0] + theta[i] * cv[1];
            }
            double pv_ll = orc_ll_bar(f + j * n, y + j * n, mu_p, n);          /* :27 */
            double cv_ll = orc_ll_bar(f + j * n, y + j * n, mu_c, n);          /* :28 */
            double rr = pv_prior + pv_ll - cv_prior - cv_ll;                   /* :29 */
            if (r->kind == ORC_RNG_ITEM) r->index = (uint32_t)(2 * k + 1);
            if (log(orc_runif(r, 0.0, 1.0)) < rr) cv[k] = pv[k]; else pv[k] = cv[k]; /* :30-35 */
        }
        beta_out[0 + 2 * j] = cv[0];
        beta_out[1 + 2 * j] = cv[1];
    }
    free(mu_p); free(mu_c);
}

/* mu = X * beta, X = [1, x]: src/gpirtMCMC.cpp:33,40,74-75,93-94 */
static void linear_mean(const double* x, int64_t n, const double* beta, int64_t m, double* mu)
{
    for (int64_t j = 0; j < m; ++j)
        for (int64_t i = 0; i < n; ++i)
            mu[i + j * n] = 1.0 * beta[0 + 2 * j] + x[i] * beta[1 + 2 * j];
}

/* src/gpirtMCMC.cpp:5-117 */
int orc_gpirt_mcmc(orc_rng* r, const double* y, int64_t n, int64_t m, const double* theta0,
                   int S_it, int B_it, const double* pm, const double* ps, const double* step,
                   const orc_mcmc_opts* opts, double* theta_draws, double* beta_draws,
                   double* f_draws, double* IRFs, double* L_final, double* fstar_final)
{
    const int64_t N = ORC_NGRID;
    orc_mcmc_opts o = { 0, 0, 1, 0 };
    if (opts) o = *opts;
    int info = 0;
    double* theta   = (double*)malloc(sizeof(double) * (size_t)n);
    double* thetan  = (double*)malloc(sizeof(double) * (size_t)n);
    double* Lm      = (double*)malloc(sizeof(double) * (size_t)(n * n));
    double* f       = (double*)malloc(sizeof(double) * (size_t)(n * m));
    double* fn      = (double*)malloc(sizeof(double) * (size_t)(n * m));
    double* z       = (double*)malloc(sizeof(double) * (size_t)n);
    double* beta    = (double*)malloc(sizeof(double) * (size_t)(2 * m));
    double* betan   = (double*)malloc(sizeof(double) * (size_t)(2 * m));
    double* mu      = (double*)malloc(sizeof(double) * (size_t)(n * m));
    double* mu_star = (double*)malloc(sizeof(double) * (size_t)(N * m));
    double* f_star  = (double*)malloc(sizeof(double) * (size_t)(N * m));
    double* tstar   = (double*)malloc(sizeof(double) * (size_t)N);
    double* tprior  = (double*)malloc(sizeof(double) * (size_t)N);
    memcpy(theta, theta0, sizeof(double) * (size_t)n);

#define ORC_FACTOR()                                                                    \
    do {                                                                                \
        orc_se_kernel(theta, n, theta, n, Lm);                  /* :15,76,95 */         \
        for (int64_t d = 0; d < n; ++d) Lm[d + d * n] += 0.001; /* :16,77,96 */         \
        info = o.blocked_potrf ? orc_potrf_lower_blocked(Lm, n, o.nthreads)             \
                               : orc_potrf_lower(Lm, n);        /* :17,78,97 */         \
    } while (0)
#define ORC_FSTAR(it)                                                                   \
    do {                                                                                \
        if (o.fstar_fused) orc_draw_fstar_fused(r, it, f, theta, tstar, Lm, mu_star, n, m, N, f_star); \
        else orc_draw_fstar(r, it, f, theta, tstar, Lm, mu_star, n, m, N, f_star, NULL, NULL); \
    } while (0)

    ORC_FACTOR();
    if (info) goto done;
    for (int64_t j = 0; j < m; ++j) {                                            /* :18-21 */
        orc_rng_substream(r, 0, ORC_ST_INIT_F, (uint32_t)j);
        orc_rmvnorm(r, Lm, n, z, f + j * n);
    }
    for (int64_t j = 0; j < m; ++j) {                                            /* :22-27 */
        orc_rng_substream(r, 0, ORC_ST_INIT_BETA, (uint32_t)j);
        for (int p = 0; p < 2; ++p) {
            if (r->kind == ORC_RNG_ITEM) r->index = (uint32_t)p;
            beta[p + 2 * j] = orc_rnorm(r, pm[p + 2 * j], ps[p + 2 * j]);
        }
    }
    linear_mean(theta, n, beta, m, mu);                                          /* :30-33 */
    orc_theta_star_grid(tstar);                                                  /* :35 */
    linear_mean(tstar, N, beta, m, mu_star);                                     /* :37-40 */
    ORC_FSTAR(0);                                                                /* :41 */
    for (int64_t i = 0; i < N * m; ++i) IRFs[i] = 0.0;                           /* :42 */
    for (int64_t i = 0; i < N; ++i) tprior[i] = orc_dnorm_log(tstar[i], 0.0, 1.0); /* :44-47 */
    for (int64_t i = 0; i < n; ++i) theta_draws[0 + i * (S_it + 1)] = theta[i];  /* :53 */
    memcpy(beta_draws, beta, sizeof(double) * (size_t)(2 * m));                  /* :54 */
    memcpy(f_draws, f, sizeof(double) * (size_t)(n * m));                        /* :55 */

    for (int it = 0; it < B_it + S_it; ++it) {                                   /* :60-104 */
        uint32_t iter = (uint32_t)(it + 1);
        orc_draw_f(r, iter, f, y, Lm, mu, n, m, fn, NULL);                       /* :68,87 */
        { double* t = f; f = fn; fn = t; }
        ORC_FSTAR(iter);                                                         /* :69,88 */
        orc_draw_theta(r, iter, tstar, y, tprior, f_star, n, m, N, o.theta_stabilise, thetan); /* :70,89 */
        { double* t = theta; theta = thetan; thetan = t; }
        orc_draw_beta(r, iter, beta, theta, y, f, pm, ps, step, n, m, betan);    /* :71-73,90-92 */
        { double* t = beta; beta = betan; betan = t; }
        linear_mean(theta, n, beta, m, mu);                                      /* :74,93 */
        linear_mean(tstar, N, beta, m, mu_star);                                 /* :75,94 */
        ORC_FACTOR();                                                            /* :76-78,95-97 */
        if (info) goto done;
        if (it >= B_it) {
            int64_t sl = it - B_it + 1;
            for (int64_t i = 0; i < n; ++i) theta_draws[sl + i * (S_it + 1)] = theta[i]; /* :99 */
            memcpy(beta_draws + sl * 2 * m, beta, sizeof(double) * (size_t)(2 * m));     /* :100 */
            memcpy(f_draws + sl * n * m, f, sizeof(double) * (size_t)(n * m));           /* :101 */
            for (int64_t i = 0; i < N * m; ++i) IRFs[i] += f_star[i];                    /* :103 */
        }
    }
    {
        double inv = 1.0 / (double)S_it;                                         /* :106 (Q7) */
        for (int64_t i = 0; i < N * m; ++i) IRFs[i] = orc_plogis(IRFs[i] * inv); /* :107-111 */
    }
    if (L_final) memcpy(L_final, Lm, sizeof(double) * (size_t)(n * n));
    if (fstar_final) memcpy(fstar_final, f_star, sizeof(double) * (size_t)(N * m));
done:
    free(theta); free(thetan); free(Lm); free(f); free(fn); free(z); free(beta); free(betan);
    free(mu); free(mu_star); free(f_star); free(tstar); free(tprior);
    return info;
}
