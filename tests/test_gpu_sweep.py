"""Randomised shape sweep: ragged respondent counts around every blocking boundary of the HIP path
(16 / 64 / 128 / 256 / 512) and odd item counts, both RNG contracts, one iteration each, against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1), (2, 3), (15, 2), (16, 1), (17, 5), (63, 4), (64, 3), (65, 2), (127, 3), (128, 2), (129, 7),
          (191, 2), (193, 3), (255, 2), (256, 3), (257, 2), (300, 65), (511, 2), (512, 3), (513, 2), (640, 5), (777, 9)]


@pytest.mark.parametrize("n,m", SHAPES)
def test_shape_sweep_item_rng(handle, oracle, n, m):
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.synthetic import make_responses
    if n < 3:
        y = np.array([[1.0] * m, [-1.0] * m])[:n] if n == 2 else np.array([[1.0] * m])
        y = np.asfortranarray(y)
        th0 = np.linspace(-1, 1, n)
    else:
        y, th0 = make_responses(n, m, seed=n * 31 + m)
    codes = dict(yea=[1], nay=[-1], missing=[None])
    from gpirt_amd.response_matrix import ResponseMatrix
    yy = np.asfortranarray(y).view(ResponseMatrix)          # keep unanimous columns (tiny n)
    res = gpirtMCMC(yy, 1, 1, vote_codes=codes, theta_init=th0, rng="item", seed=n + m, theta_stabilise=True)
    ref = oracle.gpirt_mcmc(oracle.ItemStream(n + m), np.asarray(y), th0, 1, 1, theta_stabilise=True)
    assert np.array_equal(res["theta"], ref["theta"])
    assert np.abs(res["f"] - ref["f"]).max() <= 1e-9
    assert np.abs(res["beta"] - ref["beta"]).max() <= 1e-9
    assert np.abs(res["IRFs"] - ref["IRFs"]).max() <= 1e-9


@pytest.mark.parametrize("n,m", [(17, 3), (64, 2), (129, 4), (200, 6), (257, 3)])
def test_shape_sweep_reference_rng(handle, oracle, n, m):
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    y, _ = make_responses(n, m, seed=n * 7 + m, snap_theta=False)
    rs = RStream(n)
    res = gpirtMCMC(y, 1, 1, vote_codes=dict(yea=[1], nay=[-1], missing=[None]), rng="reference", rstream=rs)
    r = oracle.RStream(n)
    ref = oracle.gpirt_mcmc(r, y, r.rnorm(n), 1, 1)
    assert np.array_equal(res["theta"], ref["theta"])
    assert np.abs(res["f"] - ref["f"]).max() <= 1e-9
    assert np.abs(res["IRFs"] - ref["IRFs"]).max() <= 1e-9
    assert rs.state()[1] == r.mt_state()[1] and np.array_equal(rs.state()[0], r.mt_state()[0])
