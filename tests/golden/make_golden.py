"""Generate the golden fixtures of the hot path (run in the build container; output is committed).

The reference ships no golden vectors and cannot run here (no R), so the vectors come from the C
restatement (oracle/gpirt_oracle.c) and are accepted only where the independent NumPy/SciPy-LAPACK
statement (oracle/np_oracle.py) agrees to 1e-10 -- see oracle/gpirt_oracle.h ("parity unpinned").

Per case (n, m, seed): K, L, the first ess() of a draw_f (z, nu, u, eps0, eps_final, k, log_y, f'),
draw_f rejection counts, draw_fstar (s, mean, f*), and a short full gpirtMCMC run under R's stream
(theta/beta/f draws, IRFs, number of uniforms consumed, final Mersenne-Twister position).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O, np_oracle as NP  # noqa: E402
from gpirt_amd.synthetic import make_responses  # noqa: E402

CASES = [(8, 3, 1234), (100, 16, 1119)]

for n, m, seed in CASES:
    y, _ = make_responses(n, m, seed=seed, snap_theta=False)
    r = O.RStream(seed)
    theta0 = r.rnorm(n)
    K = O.se_kernel(theta0, theta0)
    L, info = O.factor(theta0)
    assert info == 0
    Lnp = NP.factor(theta0)
    assert np.abs(L - Lnp).max() < 1e-10
    # one ess() with traces, from a fresh stream position
    r1 = O.RStream(seed + 1)
    f0 = L @ np.array([r1.rnorm(n) for _ in range(m)]).T
    beta = np.vstack([np.linspace(-1, 1, m), np.linspace(0.5, 2, m)])
    mu = beta[0][None, :] + theta0[:, None] * beta[1][None, :]
    r2 = O.RStream(seed + 2)
    fp, nu, tr = O.ess(r2, f0[:, 0], y[:, 0], L, mu[:, 0])
    r2b = NP.RStreamNP(seed + 2)
    fp_np, k_np = NP.ess(r2b, f0[:, 0].copy(), np.array(y[:, 0]), Lnp, mu[:, 0].copy())
    assert k_np == tr["k"] and np.abs(fp - fp_np).max() < 1e-10
    r3 = O.RStream(seed + 3)
    fnew, ks = O.draw_f(r3, f0, y, L, mu)
    ts = O.theta_star()
    mu_star = beta[0][None, :] + ts[:, None] * beta[1][None, :]
    r4 = O.RStream(seed + 4)
    fstar, s, mean = O.draw_fstar(r4, fnew, theta0, L, mu_star)
    r4b = NP.RStreamNP(seed + 4)
    fstar_np, s_np, mean_np = NP.draw_fstar(r4b, fnew, theta0, ts, Lnp, mu_star)
    assert np.abs(fstar - fstar_np).max() < 1e-9 and np.abs(s - s_np).max() < 1e-10
    # full run under R's stream
    S, B = 2, 1
    rr = O.RStream(seed)
    th = rr.rnorm(n)
    res = O.gpirt_mcmc(rr, y, th, S, B)
    rn = NP.RStreamNP(seed)
    thn = np.array([rn.rnorm(0, 1) for _ in range(n)])
    pm, ps, st = np.zeros((2, m)), np.full((2, m), 3.0), np.full((2, m), 0.1)
    resn = NP.gpirt_mcmc(rn, np.array(y), thn, S, B, pm, ps, st)
    for key in ("theta", "beta", "f", "IRFs"):
        assert np.abs(res[key] - resn[key]).max() < 1e-9, key
    assert rr.n_unif == rn.n_unif
    mt, mti = rr.mt_state()
    yenc = np.where(np.isnan(y), 0, y).astype(np.int8)
    np.savez_compressed(
        os.path.join(HERE, f"hotpath_n{n}_m{m}.npz"), seed=seed, y=yenc, theta0=theta0, K=K, L=L, f0=f0, mu=mu,
        ess_fprime=fp, ess_nu=nu, ess_u=tr["u"], ess_log_y=tr["log_y"], ess_eps0=tr["eps0"],
        ess_eps_final=tr["eps_final"], ess_k=tr["k"], drawf_out=fnew, drawf_k=ks, mu_star=mu_star,
        fstar=fstar, fstar_s=s, fstar_mean=mean, mcmc_theta=res["theta"], mcmc_beta=res["beta"], mcmc_f=res["f"],
        mcmc_irfs=res["IRFs"], mcmc_n_unif=rr.n_unif, mcmc_mti=mti, mcmc_mt_head=mt[:8])
    print("wrote case", n, m, "uniforms consumed", rr.n_unif)
