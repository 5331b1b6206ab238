"""The C restatement against (i) the committed golden fixtures and (ii) the independent
NumPy/SciPy-LAPACK statement, stage by stage, plus the reference quirks SURVEY.md 7.3-H2 lists."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["hotpath_n8_m3.npz", "hotpath_n100_m16.npz"]


def _load(name):
    d = np.load(os.path.join(GOLD, name))
    y = d["y"].astype(np.float64)
    y[y == 0] = np.nan
    return d, y


@pytest.mark.parametrize("name", CASES)
def test_golden_kernel_and_cholesky(oracle, name):
    d, y = _load(name)
    th = d["theta0"]
    assert np.array_equal(oracle.RStream(int(d["seed"])).rnorm(len(th)), th)
    assert np.abs(oracle.se_kernel(th, th) - d["K"]).max() == 0
    L, info = oracle.factor(th)
    assert info == 0 and np.abs(L - d["L"]).max() == 0
    Lb, info = oracle.factor(th, blocked=True, nthreads=2)
    assert info == 0 and np.abs(Lb - d["L"]).max() < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_golden_ess_trace(oracle, name):
    d, y = _load(name)
    seed = int(d["seed"])
    fp, nu, tr = oracle.ess(oracle.RStream(seed + 2), d["f0"][:, 0], y[:, 0], d["L"], d["mu"][:, 0])
    assert tr["k"] == int(d["ess_k"])
    for key in ("u", "log_y", "eps0", "eps_final"):
        assert tr[key] == float(d["ess_" + key])
    assert np.array_equal(fp, d["ess_fprime"]) and np.array_equal(nu, d["ess_nu"])
    # quirk Q1: eps_min = eps0 - 2 pi but eps_max stays 2 pi, so a later eps may exceed eps0
    assert -2 * np.pi < tr["eps_final"] < 2 * np.pi


@pytest.mark.parametrize("name", CASES)
def test_golden_draw_f_and_fstar(oracle, name):
    d, y = _load(name)
    seed = int(d["seed"])
    out, ks = oracle.draw_f(oracle.RStream(seed + 3), d["f0"], y, d["L"], d["mu"])
    assert np.array_equal(ks, d["drawf_k"]) and np.array_equal(out, d["drawf_out"])
    fstar, s, mean = oracle.draw_fstar(oracle.RStream(seed + 4), d["drawf_out"], d["theta0"], d["L"], d["mu_star"])
    assert np.array_equal(fstar, d["fstar"]) and np.array_equal(s, d["fstar_s"])
    # quirk Q2: s = 1 - sqrt(q) is tiny and positive
    assert np.all(s > 0) and s.max() < 1.0


@pytest.mark.parametrize("name", CASES)
def test_golden_full_mcmc(oracle, name):
    d, y = _load(name)
    r = oracle.RStream(int(d["seed"]))
    th = r.rnorm(y.shape[0])
    res = oracle.gpirt_mcmc(r, y, th, 2, 1)
    assert np.array_equal(res["theta"], d["mcmc_theta"])
    assert np.array_equal(res["beta"], d["mcmc_beta"])
    assert np.array_equal(res["f"], d["mcmc_f"])
    assert np.array_equal(res["IRFs"], d["mcmc_irfs"])
    assert r.n_unif == int(d["mcmc_n_unif"])
    mt, mti = r.mt_state()
    assert mti == int(d["mcmc_mti"]) and np.array_equal(mt[:8], d["mcmc_mt_head"])
    # after the first iteration every theta lies on the -5:0.01:5 grid (Q6); slot 0 is theta_init
    k = (res["theta"][1:] + 5.0) / 0.01
    assert np.abs(k - np.rint(k)).max() < 1e-9


def test_c_oracle_vs_numpy_statement(oracle):
    from oracle import np_oracle as NP
    from gpirt_amd.synthetic import make_responses
    n, m = 36, 5
    y, _ = make_responses(n, m, seed=3, snap_theta=False)
    r = oracle.RStream(714)
    th0 = r.rnorm(n)
    a = oracle.gpirt_mcmc(r, y, th0, 2, 2, want_state=True)
    rn = NP.RStreamNP(714)
    th0n = np.array([rn.rnorm(0, 1) for _ in range(n)])
    pm, ps, st = np.zeros((2, m)), np.full((2, m), 3.0), np.full((2, m), 0.1)
    b = NP.gpirt_mcmc(rn, np.array(y), th0n, 2, 2, pm, ps, st)
    assert r.n_unif == rn.n_unif
    for key in ("theta", "beta", "f", "IRFs", "L", "fstar"):
        assert np.nanmax(np.abs(a[key] - b[key])) < 1e-9, key


def test_blocked_potrf_matches_unblocked_and_lapack(oracle):
    import scipy.linalg as sla
    n = 700
    rng = np.random.default_rng(0)
    th = -5 + 0.01 * rng.integers(0, 1001, n)
    S = oracle.se_kernel(th, th)
    S[np.diag_indices(n)] += 0.001
    Lu, i1 = oracle.potrf_lower(S)
    Lb, i2 = oracle.potrf_lower(S, blocked=True, nthreads=4)
    Ll = sla.cholesky(S, lower=True)
    assert i1 == 0 and i2 == 0
    assert np.abs(Lu - Ll).max() < 1e-11 and np.abs(Lb - Ll).max() < 1e-11
    assert np.array_equal(np.triu(Lb, 1), np.zeros_like(Lb))


def test_potrf_reports_leading_minor(oracle):
    S = np.eye(50)
    S[30, 30] = -2.0
    for blocked in (False, True):
        _, info = oracle.potrf_lower(S, blocked=blocked, nthreads=1)
        assert info == 31


def test_ll_skips_nan_and_matches_formula(oracle):
    f = np.array([0.3, -1.2, 2.0, 0.7])
    y = np.array([1.0, np.nan, -1.0, 1.0])
    mu = np.array([0.1, 0.2, -0.3, 0.0])
    want = -(np.log(1 + np.exp(-(0.4))) + np.log(1 + np.exp(1.7)) + np.log(1 + np.exp(-0.7)))
    assert abs(oracle.ll_bar(f, y, mu) - want) < 1e-15
    assert abs(oracle.ll(f + mu, y) - want) < 1e-15


def test_draw_theta_quirks(oracle):
    from gpirt_amd.synthetic import make_responses
    n, m = 30, 8
    y, _ = make_responses(n, m, seed=5)
    ts = oracle.theta_star()
    fstar = 0.5 + 1.5 * ts[:, None] * np.ones((1, m))
    th, deg = oracle.draw_theta(oracle.ItemStream(1), y, fstar, it=1)
    ths, degs = oracle.draw_theta(oracle.ItemStream(1), y, fstar, it=1, stabilise=True)
    assert deg == 0 and degs == 0 and np.array_equal(th, ths)
    assert np.all(th > ts[0])                       # Q5: grid point 0 is never selectable
    # underflow of exp() makes the reference's CDF degenerate (0/0): reported, NaN returned
    big = np.tile(fstar, (1, 300))
    ybig = np.tile(y, (1, 300))
    th, deg = oracle.draw_theta(oracle.ItemStream(1), ybig, big * 4, it=1)
    ths, degs = oracle.draw_theta(oracle.ItemStream(1), ybig, big * 4, it=1, stabilise=True)
    assert deg > 0 and np.isnan(th).sum() == deg
    assert degs == 0 and not np.isnan(ths).any()


def test_sample_iterations_zero_gives_nan_irfs(oracle):
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(12, 3, seed=2)
    with np.errstate(all="ignore"):
        res = oracle.gpirt_mcmc(oracle.RStream(1), y, th0, 0, 1)
    assert np.isnan(res["IRFs"]).all()              # Q7
    assert res["theta"].shape == (1, 12)


def test_fstar_fused_form_agrees(oracle):
    from gpirt_amd.synthetic import make_responses
    n, m = 120, 4
    y, th0 = make_responses(n, m, seed=8)
    a = oracle.gpirt_mcmc(oracle.ItemStream(3), y, th0, 1, 1, theta_stabilise=True)
    b = oracle.gpirt_mcmc(oracle.ItemStream(3), y, th0, 1, 1, theta_stabilise=True, fstar_fused=True)
    assert np.array_equal(a["theta"], b["theta"])
    assert np.abs(a["IRFs"] - b["IRFs"]).max() < 1e-9


def test_senate116_fixture_shape():
    d = np.load(os.path.join(GOLD, "senate116_y.npz"))
    y = d["y"]
    assert y.shape == (100, 418) and set(np.unique(y)) == {-1, 0, 1}
    assert abs((y == 0).mean() - 0.0574) < 1e-3


def test_cpu_baseline_leg_runs_on_a_small_problem(oracle):
    """bench.py's cpu_baseline leg (oracle/cpu_baseline.py): both legs produce finite, positive rates and say how far
    each sample was stretched."""
    from oracle import cpu_baseline as CB
    from gpirt_amd.synthetic import make_responses
    n, m = 256, 24
    y, th = make_responses(n, m, seed=3)
    L, info = oracle.factor(th)
    assert info == 0
    rng = np.random.default_rng(0)
    f = np.asfortranarray(L @ rng.standard_normal((n, m)))
    beta = np.asfortranarray(rng.standard_normal((2, m)))
    mu = np.asfortranarray(beta[0][None, :] + th[:, None] * beta[1][None, :])
    fstar = np.asfortranarray(rng.standard_normal((1001, m)))
    r = CB.run(n, m, y, th, L, f, beta, mu, fstar, nthreads=2)
    # the top-level figure is the MEASURED one: one whole iteration at the full size on all the cores asked for
    assert r["kind"] == "port" and r["cores"] == 2 and r["extrapolated"] is False and r["value"] > 0 and np.isfinite(r["value"])
    a = r["single_thread_reference_shaped"]                      # the reference-shaped single-thread leg, stretched from samples
    assert a["cores"] == 1 and a["value"] > 0 and np.isfinite(a["value"]) and a["extrapolated"] is True
    assert set(a["stage_seconds"]) == {"K", "chol", "draw_f", "draw_fstar", "draw_theta", "draw_beta"}
    assert set(r["stage_seconds"]) == {"K", "chol", "draw_f", "draw_fstar", "draw_theta", "draw_beta"}
    assert a["extrapolation_factors"]["all.chol"] == 1.0      # the all-core potrf runs at the full size
