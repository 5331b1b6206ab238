"""Known-answer tests that pin the oracle's (and the product's host-side) restatement of R's RNG.

The values are public R facts: set.seed(s); runif(3) / rnorm(3) for the seeds
the reference's own examples use or that every R user knows, qnorm at two points, and the published
Random123 vectors for Philox4x32-10 (the item-RNG contract).
"""
import numpy as np
import pytest


R_RUNIF = {
    42: [0.9148060434963554, 0.9370754132978618, 0.2861395347863436],
    1: [0.2655087, 0.3721239, 0.5728534],
    123: [0.2875775, 0.7883051, 0.4089769],
}
R_RNORM = {
    42: [1.37095845, -0.56469817, 0.36312841],
    1: [-0.6264538, 0.1836433, -0.8356286],
    123: [-0.56047565, -0.23017749, 1.55870831],
    1234: [-1.2070657, 0.2774292, 1.0844412, -2.3456977, 0.4291247],
}


@pytest.mark.parametrize("seed", sorted(R_RUNIF))
def test_runif_kat(oracle, seed):
    got = oracle.RStream(seed).runif(3)
    assert np.allclose(got, R_RUNIF[seed], rtol=0, atol=5e-8)
    if seed == 42:
        assert np.array_equal(got, R_RUNIF[42])


@pytest.mark.parametrize("seed", sorted(R_RNORM))
def test_rnorm_kat(oracle, seed):
    ref = R_RNORM[seed]
    got = oracle.RStream(seed).rnorm(len(ref))
    assert np.allclose(got, ref, rtol=0, atol=5e-8)


def test_qnorm_kat(oracle):
    assert abs(oracle.qnorm(0.975) - 1.959963984540054) < 2e-15
    assert abs(oracle.qnorm(0.1) + 1.2815515655446008) < 2e-15
    assert oracle.qnorm(0.5) == 0.0
    from scipy.special import ndtri
    p = np.concatenate([np.linspace(1e-12, 1 - 1e-12, 2001), 10.0 ** -np.arange(1, 18)])
    q = np.array([oracle.qnorm(v) for v in p])
    assert np.max(np.abs(q - ndtri(p)) / np.maximum(1.0, np.abs(q))) < 5e-15


def test_mt_core_matches_numpy(oracle):
    """The hand-written Mersenne-Twister equals numpy's MT19937 core under R's seeding."""
    from oracle.np_oracle import RStreamNP
    a, b = oracle.RStream(1119), RStreamNP(1119)
    xs = a.runif(2000)
    ys = np.array([b.unif_rand() for _ in range(2000)])
    assert np.array_equal(xs, ys)


def test_rnorm_sd_zero_consumes_nothing(oracle):
    r = oracle.RStream(7)
    lib = oracle.lib()
    assert lib.orc_rnorm(r.ref, 2.5, 0.0) == 2.5 and r.n_unif == 0
    assert np.isnan(lib.orc_rnorm(r.ref, 0.0, -1.0)) and r.n_unif == 0
    lib.orc_rnorm(r.ref, 0.0, 1.0)
    assert r.n_unif == 2


def test_philox_random123_vectors(oracle):
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_item_uniform_open_interval(oracle):
    us = np.array([oracle.item_uniform(99, 1, 3, j, i) for j in range(20) for i in range(50)])
    assert np.all((us > 0) & (us < 1)) and len(np.unique(us)) == len(us)
    assert abs(us.mean() - 0.5) < 0.05


def test_product_host_rstream_matches_oracle(oracle):
    """gpirt_rstream_* in libgpirt_hip.so (host code, no GPU needed) is the same generator."""
    from gpirt_amd.ops import RStream
    a, b = RStream(1234), oracle.RStream(1234)
    assert np.array_equal(a.rnorm(500), b.rnorm(500))
    assert np.array_equal(a.runif(500), b.runif(500))
    mt, mti = a.state()
    mtb, mtib = b.mt_state()
    assert mti == mtib and np.array_equal(mt, mtb)
    c = RStream(state=(mt, mti))
    assert np.array_equal(c.runif(10), b.runif(10))
