"""log(1 + exp(-a)) of the elliptical-slice kernel (gpirt_amd/csrc/ll_fast.h).

CPU: the header compiles for the host too, so its arithmetic is checked here against long double on 4 M points (the
device differs from this build only in the seed of the reciprocal, which two Newton steps wash out).  GPU: the device
function itself through gpirt_debug_ll_term against numpy long double, and the slice sampler with the written form
(GPIRT_LL_EXACT=1) against the default on the same inputs: same rejection counts, draws equal to rounding."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HOST_PROG = r"""
#include "ll_fast.h"
#include <stdio.h>
#include <stdlib.h>
int main() {
    double worst = 0, worst_a = 0, worst_abs = 0; long n = 0;
    srand(7);
    auto check = [&](double a) {
        long double ref = (a < 0 ? -(long double)a : 0.0L) + log1pl(expl(-fabsl((long double)a)));
        double got = ll_term_fast(a);
        double ulp = nextafter((double)ref, INFINITY) - (double)ref;
        double err = fabs((double)((long double)got - ref)) / ulp;
        if (err > worst) { worst = err; worst_a = a; }
        if (a >= 0) { double d = fabs((double)((long double)got - ref)); if (d > worst_abs) worst_abs = d; }
        ++n;
    };
    for (double a = -60; a <= 60; a += 0.00037) check(a);
    for (int i = 0; i < 2000000; ++i) check(-745.0 + 1490.0 * rand() / RAND_MAX);
    for (int i = 0; i < 2000000; ++i) check(-3.0 + 6.0 * rand() / RAND_MAX);
    printf("%ld %.4f %.17g %.4e\n", n, worst, worst_a, worst_abs);
    printf("%g %g %d %g %.17g\n", ll_term_fast(INFINITY), ll_term_fast(-INFINITY), (int)(ll_term_fast(NAN) != ll_term_fast(NAN)),
           ll_term_fast(-800.0), ll_term_fast(0.0));
    return 0;
}
"""


def test_host_build_of_the_header_is_within_3_ulp_of_long_double():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(HOST_PROG)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "gpirt_amd", "csrc"), src, "-o", exe, "-lm"])
        out = subprocess.check_output([exe]).decode().split("\n")
    n, worst, worst_a, worst_abs = out[0].split()
    assert int(n) > 4_000_000
    assert float(worst) < 3.0, (worst, worst_a)            # ulp of the exact value (the written form is ~2.5 ulp off it itself)
    assert float(worst_abs) < 2.0e-16                        # a >= 0: absolute error of a term of magnitude <= log 2
    sp = out[1].split()
    assert sp[0] == "0" and sp[1] == "inf" and sp[2] == "1" and float(sp[3]) == 800.0 and abs(float(sp[4]) - np.log(2.0)) < 2e-16


@pytest.mark.gpu
def test_device_function_against_long_double():
    import torch
    from gpirt_amd.ops import Handle
    h = Handle()
    rng = np.random.default_rng(5)
    a = np.concatenate([np.linspace(-60, 60, 400001), rng.uniform(-745, 745, 400000), rng.uniform(-3, 3, 400000),
                        [0.0, -0.0, 37.0, 40.0, 745.2, -745.2, -800.0, 1e-300, -1e-300]])
    al = a.astype(np.longdouble)
    ref = np.where(al < 0, -al, 0) + np.log1p(np.exp(-np.abs(al)))
    got = h.ll_term(torch.from_numpy(a).cuda(), fast=True).cpu().numpy()
    ulp = np.spacing(ref.astype(np.float64))
    err = np.abs((got.astype(np.longdouble) - ref).astype(np.float64)) / ulp
    assert err.max() < 3.0, (err.max(), a[err.argmax()])
    # and the written form, for scale: it rounds 1 + exp(-a) first
    wr = h.ll_term(torch.from_numpy(a).cuda(), fast=False).cpu().numpy()
    fin = np.isfinite(wr)
    assert np.abs(wr[fin] - got[fin]).max() < 2e-13          # (|a| up to 745: an ulp of the value there is 1.1e-13)
    pos = fin & (a >= 0)
    assert np.abs(wr[pos] - got[pos]).max() < 2.5e-16
    sp = h.ll_term(torch.tensor([float("inf"), float("-inf"), float("nan")], dtype=torch.float64).cuda()).cpu().numpy()
    assert sp[0] == 0.0 and np.isinf(sp[1]) and np.isnan(sp[2])


@pytest.mark.gpu
def test_slice_sampler_with_the_written_form_agrees():
    """draw_f on the same state and RNG keys with GPIRT_LL_EXACT=1 (library exp and log, the formula as written) and with
    the default: identical rejection counts, draws equal to rounding (a decision flips only if a log-likelihood sum lands
    within ~1e-12 of the slice level)."""
    import torch
    from gpirt_amd.ops import Handle
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 1024, 96
    y, th0 = make_responses(n, m, seed=31)
    h = Handle()
    outs = []
    for exact in (0, 1):
        with h.config("GPIRT_LL_EXACT", exact):
            s = Sampler(h, y, th0, rng="item", seed=77, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
            s.init()
            for _ in range(3):
                s.step()
            outs.append((s.get("f"), s.get("ess_k"), s.get("theta")))
            s.close()
    (f0, k0, t0), (f1, k1, t1) = outs
    assert np.array_equal(k0, k1)
    assert np.array_equal(t0, t1)
    assert np.abs(f0 - f1).max() < 1e-10 * max(1.0, np.abs(f1).max())


@pytest.mark.gpu
def test_screen_of_the_accept_test_stays_inside_its_error_bound():
    """ll_term_screen (single precision through v_exp_f32 / v_log_f32) against long double: the bound the slice kernel
    widens its decisions by is LL_SCREEN_ERR = 4e-6 per row (csrc/ll_fast.h); the measured maximum has to stay under a
    quarter of it, at every magnitude (the part max(-a, 0) of a term is carried in fp64, so the error does not grow with |a|)."""
    import re
    import torch
    from gpirt_amd.ops import Handle
    hdr = open(os.path.join(ROOT, "gpirt_amd", "csrc", "ll_fast.h")).read()
    bound = float(re.search(r"#define LL_SCREEN_ERR ([0-9.eE+-]+)", hdr).group(1))
    h = Handle()
    rng = np.random.default_rng(6)
    a = np.concatenate([np.linspace(-60, 60, 1200001), rng.uniform(-745, 745, 1000000), rng.uniform(-3, 3, 1000000),
                        rng.uniform(-20, 20, 1000000), rng.standard_normal(200000) * 1e-3,
                        [0.0, -0.0, 37.0, 40.0, 87.0, 88.8, 104.0, 745.2, -745.2, -800.0, 1e-300, -1e-300, 1e30, -1e30, 1e300, -1e300]])
    al = a.astype(np.longdouble)
    ref = np.where(al < 0, -al, 0) + np.log1p(np.exp(-np.abs(al)))
    got = h.ll_term(torch.from_numpy(a).cuda(), screen=True).cpu().numpy()
    err = np.abs((got.astype(np.longdouble) - ref).astype(np.float64))
    assert err.max() < bound / 4, (err.max(), a[err.argmax()])
    sp = h.ll_term(torch.tensor([float("inf"), float("-inf"), float("nan")], dtype=torch.float64).cuda(), screen=True).cpu().numpy()
    assert sp[0] == 0.0 and np.isinf(sp[1]) and sp[1] > 0 and np.isnan(sp[2])


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(1024, 96), (2048, 64), (4096, 48), (8192, 40)])
def test_slice_sampler_with_and_without_the_screen_is_bit_identical(n, m):
    """The screen only decides what the full-precision sum would decide the same way: with GPIRT_ESS_SCREEN=2 (every trial
    point in full precision) the chain is the same bit for bit -- rejection counts, f, theta -- in both register kernels
    (n <= 2048: 8 rows per thread; above: 16) and with the written form of the term (GPIRT_LL_EXACT=1) as the referee too."""
    import torch
    from gpirt_amd.ops import Handle
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(n, m, seed=41)                   # (5% missing responses: skipped by both passes alike)
    h = Handle()
    for exact in (0, 1):
        outs = []
        for screen in (1, 2):
            with h.config("GPIRT_LL_EXACT", exact), h.config("GPIRT_ESS_SCREEN", screen):
                s = Sampler(h, y, th0, rng="item", seed=78, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
                s.init()
                for _ in range(3):
                    s.step()
                outs.append((s.get("f"), s.get("ess_k"), s.get("theta")))
                s.close()
        (f0, k0, t0), (f1, k1, t1) = outs
        assert k0.sum() > m                                  # (there were rejections to decide)
        assert np.array_equal(k0, k1)
        assert np.array_equal(f0, f1)
        assert np.array_equal(t0, t1)
