"""The all-core oracle drivers (tests/_oracle_parallel.py) reproduce the sequential oracle functions bit for bit:
slices of the item / grid-column / respondent loops carry the global index into the RNG keys."""
import numpy as np

from tests import _oracle_parallel as P


def test_slices_reproduce_the_sequential_stages(oracle):
    O = oracle
    from gpirt_amd.synthetic import make_responses
    n, m, seed, it = 96, 23, 5, 3
    y, th0 = make_responses(n, m, seed=20250, snap_theta=True)
    rng = np.random.default_rng(1)
    L, info = O.factor(th0)
    assert info == 0
    f = np.asfortranarray(rng.normal(size=(n, m)))
    beta = np.asfortranarray(rng.normal(size=(2, m)))
    mu = np.asfortranarray(beta[0][None, :] + th0[:, None] * beta[1][None, :])
    ts = O.theta_star()
    mu_star = np.asfortranarray(beta[0][None, :] + ts[:, None] * beta[1][None, :])
    pm, ps, st = np.zeros((2, m)), np.full((2, m), 3.0), np.full((2, m), 0.1)

    f1, k1 = O.draw_f(O.ItemStream(seed), f, y, L, mu, it=it)
    f2, k2 = P.draw_f(seed, it, f, y, L, mu, nthreads=3)
    assert np.array_equal(f1, f2) and np.array_equal(k1, k2)

    fs1, s1, mean1 = O.draw_fstar(O.ItemStream(seed), f1, th0, L, mu_star, it=it)
    fs2, s2, mean2 = P.draw_fstar(seed, it, f1, th0, L, mu_star, nthreads=3)
    assert np.array_equal(s1, s2) and np.array_equal(mean1, mean2) and np.array_equal(fs1, fs2)

    t1, d1 = O.draw_theta(O.ItemStream(seed), y, fs1, it=it, stabilise=True)
    t2, d2 = P.draw_theta(seed, it, y, fs1, stabilise=True, nthreads=3)
    assert np.array_equal(t1, t2) and d1 == d2 == 0

    b1 = O.draw_beta(O.ItemStream(seed), beta, t1, y, f1, pm, ps, st, it=it)
    b2 = P.draw_beta(seed, it, beta, t1, y, f1, pm, ps, st, nthreads=3)
    assert np.array_equal(b1, b2)

    # an item shard keyed by its first global item (what a rank of an item-sharded run draws)
    f3, k3 = P.draw_f(seed, it, f[:, 7:], y[:, 7:], L, mu[:, 7:], nthreads=2, item0=7)
    assert np.array_equal(f3, f1[:, 7:]) and np.array_equal(k3, k1[7:])

    L2, info2 = P.factor(th0, nthreads=2)
    assert info2 == 0 and np.abs(np.tril(L2) - np.tril(L)).max() < 1e-12
