"""Second, independent CPU statement of the gpirt sampler in NumPy/SciPy (TEST INFRASTRUCTURE ONLY).

Purpose: cross-check the C restatement (oracle/gpirt_oracle.c) with different code and with the
LAPACK/BLAS routines the reference reaches through Armadillo:
  arma::chol(S,"lower")        -> LAPACK dpotrf  (scipy.linalg.cholesky)
  arma::solve(trimatl/u, .)    -> LAPACK dtrtrs  (scipy.linalg.solve_triangular)
  cholS * res, kstarT * alpha  -> BLAS dgemv     (numpy @)
and with NumPy's own MT19937 core (np.random.MT19937.random_raw) under R's seeding/tempering
conventions, and scipy.special.ndtri as an independent normal quantile.

Each function cites the reference file:line it restates (paths relative to the upstream repo).
Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
from scipy.special import ndtri

NGRID = 1001
TWO_PI = 6.283185307179586476925286766559


class RStreamNP:
    """R's default RNG (Mersenne-Twister, inversion) on top of numpy's MT19937 core."""

    def __init__(self, seed: int):
        s = np.uint32(seed & 0xFFFFFFFF)
        with np.errstate(over="ignore"):
            for _ in range(50):
                s = np.uint32(69069) * s + np.uint32(1)
            dummy = np.empty(625, dtype=np.uint32)
            for j in range(625):
                s = np.uint32(69069) * s + np.uint32(1)
                dummy[j] = s
        self.bg = np.random.MT19937()
        self.bg.state = {"bit_generator": "MT19937", "state": {"key": dummy[1:].copy(), "pos": 624}}
        self.n_unif = 0

    def unif_rand(self) -> float:
        self.n_unif += 1
        v = float(self.bg.random_raw()) * 2.3283064365386963e-10
        i2 = 2.328306437080797e-10
        if v <= 0.0:
            return 0.5 * i2
        if 1.0 - v <= 0.0:
            return 1.0 - 0.5 * i2
        return v

    def norm_rand(self) -> float:
        big = 134217728.0
        u = self.unif_rand()
        u = float(int(big * u)) + self.unif_rand()
        return float(ndtri(u / big))

    def rnorm(self, mu: float, sd: float) -> float:
        if np.isnan(mu) or not np.isfinite(sd) or sd < 0:
            return float("nan")
        if sd == 0.0 or not np.isfinite(mu):
            return mu
        return mu + sd * self.norm_rand()

    def runif(self, a: float, b: float) -> float:
        if a == b:
            return a
        return a + (b - a) * self.unif_rand()


def theta_star():
    return -5.0 + np.arange(NGRID) * 0.01          # src/gpirtMCMC.cpp:35


def se_kernel(x1, x2):                              # src/covariance-function.cpp:3-14
    d = np.asarray(x1)[:, None] - np.asarray(x2)[None, :]
    return np.exp(-0.5 * d * d)


def factor(theta):                                  # src/gpirtMCMC.cpp:15-17
    S = se_kernel(theta, theta)
    S[np.diag_indices_from(S)] += 0.001
    return sla.cholesky(S, lower=True)


def rmvnorm(rng, L):                                # src/mvnormal.h:4-11
    z = np.array([rng.rnorm(0.0, 1.0) for _ in range(L.shape[0])])
    return L @ z


def ll(f, y):                                       # src/log-likelihood.cpp:12-23
    ok = ~np.isnan(y)
    return -float(np.sum(np.log(1 + np.exp(-(y[ok] * f[ok])))))


def ll_bar(f, y, mu):                               # src/log-likelihood.cpp:25-37
    return ll(f + mu, y)


def ess(rng, f, y, L, mu):                          # src/draw-f.cpp:21-60
    nu = rmvnorm(rng, L)
    u = rng.runif(0.0, 1.0)
    log_y = ll_bar(f, y, mu) + np.log(u)
    eps_min, eps_max = 0.0, TWO_PI
    eps = rng.runif(eps_min, eps_max)
    eps_min = eps - TWO_PI
    k = 0
    while True:
        fp = f * np.cos(eps) + nu * np.sin(eps)
        if ll_bar(fp, y, mu) > log_y:
            return fp, k
        if eps < 0.0:
            eps_min = eps
        else:
            eps_max = eps
        eps = rng.runif(eps_min, eps_max)
        k += 1


def draw_f(rng, f, y, L, mu):                       # src/draw-f.cpp:64-73
    out = np.empty_like(f)
    ks = []
    for j in range(f.shape[1]):
        out[:, j], k = ess(rng, f[:, j], y[:, j], L, mu[:, j])
        ks.append(k)
    return out, ks


def draw_fstar(rng, f, theta, tstar, L, mu_star):   # src/draw-fstar.cpp:10-31
    kstar = se_kernel(theta, tstar)
    tmp = sla.solve_triangular(L, kstar, lower=True)
    s = 1.0 - np.sqrt(np.sum(tmp * tmp, axis=0))
    out = np.empty((len(tstar), f.shape[1]))
    means = np.empty_like(out)
    for j in range(f.shape[1]):
        alpha = sla.solve_triangular(L.T, sla.solve_triangular(L, f[:, j], lower=True), lower=False)
        mean = kstar.T @ alpha + mu_star[:, j]
        means[:, j] = mean
        for i in range(len(tstar)):
            out[i, j] = rng.rnorm(mean[i], s[i])
    return out, s, means


def draw_theta(rng, tstar, y, prior, fstar, stabilise=False):   # src/draw-theta.cpp:3-37
    n = y.shape[0]
    out = np.empty(n)
    for i in range(n):
        ok = ~np.isnan(y[i])
        a = fstar[:, ok] * y[i, ok][None, :]
        P = prior - np.sum(np.log(1 + np.exp(-a)), axis=1)
        if stabilise:
            P = P - P.max()
        P = np.cumsum(np.exp(P))
        P = (P - P.min()) / (P.max() - P.min())
        u = rng.runif(0.0, 1.0)
        idx = np.nonzero(P > u)[0]
        out[i] = tstar[idx[0]] if len(idx) else np.nan
    return out


def dnorm_log(x, mu, sd):
    z = abs((x - mu) / sd)
    return -(0.918938533204672741780329736406 + 0.5 * z * z + np.log(sd))


def draw_beta(rng, beta, theta, y, f, pm, ps, step):            # src/draw-beta.cpp:3-41
    out = np.empty_like(beta)
    for j in range(beta.shape[1]):
        cv = beta[:, j].copy()
        pv = cv.copy()
        for k in range(2):
            pv[k] = rng.rnorm(cv[k], step[k, j])
            pvp = dnorm_log(pv[k], pm[k, j], ps[k, j])
            cvp = dnorm_log(cv[k], pm[k, j], ps[k, j])
            pvl = ll_bar(f[:, j], y[:, j], pv[0] + theta * pv[1])
            cvl = ll_bar(f[:, j], y[:, j], cv[0] + theta * cv[1])
            r = pvp + pvl - cvp - cvl
            if np.log(rng.runif(0.0, 1.0)) < r:
                cv[k] = pv[k]
            else:
                pv[k] = cv[k]
        out[:, j] = cv
    return out


def gpirt_mcmc(rng, y, theta0, S, B, pm, ps, step, stabilise=False):   # src/gpirtMCMC.cpp:5-117
    n, m = y.shape
    theta = np.array(theta0, dtype=float)
    L = factor(theta)
    f = np.empty((n, m))
    for j in range(m):
        f[:, j] = rmvnorm(rng, L)
    beta = np.empty((2, m))
    for j in range(m):
        for p in range(2):
            beta[p, j] = rng.rnorm(pm[p, j], ps[p, j])
    mu = beta[0][None, :] + theta[:, None] * beta[1][None, :]
    ts = theta_star()
    mu_star = beta[0][None, :] + ts[:, None] * beta[1][None, :]
    fstar, _, _ = draw_fstar(rng, f, theta, ts, L, mu_star)
    irf = np.zeros((NGRID, m))
    prior = np.array([dnorm_log(t, 0.0, 1.0) for t in ts])
    th_d = np.empty((S + 1, n)); be_d = np.empty((2, m, S + 1)); f_d = np.empty((n, m, S + 1))
    th_d[0] = theta; be_d[:, :, 0] = beta; f_d[:, :, 0] = f
    for it in range(B + S):
        f, _ = draw_f(rng, f, y, L, mu)
        fstar, _, _ = draw_fstar(rng, f, theta, ts, L, mu_star)
        theta = draw_theta(rng, ts, y, prior, fstar, stabilise)
        beta = draw_beta(rng, beta, theta, y, f, pm, ps, step)
        mu = beta[0][None, :] + theta[:, None] * beta[1][None, :]
        mu_star = beta[0][None, :] + ts[:, None] * beta[1][None, :]
        L = factor(theta)
        if it >= B:
            sl = it - B + 1
            th_d[sl] = theta; be_d[:, :, sl] = beta; f_d[:, :, sl] = f
            irf += fstar
    with np.errstate(divide="ignore", invalid="ignore"):
        irf = 1.0 / (1.0 + np.exp(-(irf * (1.0 / S))))
    return dict(theta=th_d, beta=be_d, f=f_d, IRFs=irf, L=L, fstar=fstar)
