"""The arithmetic of gpirt_amd/csrc/theta_fixed.hip, restated in numpy / Python integers (no GPU): what the header of that file
claims about the fixed-point form of draw_theta's log-posterior sums (src/draw-theta.cpp:15-19) --

  * a term below 2^e, scaled by 2^(54 - e) and rounded, splits into seven balanced base-256 digits that are all signed bytes
    and put the integer back together exactly;
  * the int32 accumulators of the seven digit planes cannot overflow for any number of items the int8 kernel's K allows here;
  * the two int64 halves the planes are recombined in are exact in fp64, so the result carries ONE rounding;
  * and that result is within m 2^(e - 55) + one ulp of the exact sum of the fp64 terms.

The GPU tests (tests/test_gpu_theta_fixed.py) check the kernels against this scheme's bound and against the fp64 GEMM."""
from fractions import Fraction

import numpy as np

BITS, DIGITS = 54, 7


def quantise(term, e):
    return np.rint(np.ldexp(term, BITS - e)).astype(np.int64)


def digits_of(q):
    out = []
    q = q.copy()
    for s in range(DIGITS):
        if s + 1 < DIGITS:
            d = ((q & 255) ^ 128) - 128                      # the low byte read as a signed byte
        else:
            d = q.copy()
        out.append(d)
        q = (q - d) >> 8
    return out


def row_exponent(fmax):
    return int(np.floor(np.log2(fmax + 0.6931471805599453))) + 1      # ilogb(fmax + log 2) + 1


def test_digits_are_signed_bytes_and_put_the_integer_back_together():
    rng = np.random.default_rng(1)
    for fmax in (1e-3, 0.3, 1.0, 7.9, 8.0, 100.0, 709.0):
        e = row_exponent(fmax)
        # terms of a row: log(1 + exp(x)) for |x| <= fmax, the largest of them included
        x = np.concatenate([rng.uniform(-fmax, fmax, 20000), [fmax, -fmax, 0.0]])
        term = np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0.0)
        assert term.max() < 2.0 ** e
        q = quantise(term, e)
        assert q.min() >= 0 and q.max() <= 2 ** BITS
        ds = digits_of(q)
        for d in ds:
            assert d.min() >= -128 and d.max() <= 127
        back = sum(d.astype(object) * 256 ** s for s, d in enumerate(ds))
        assert (back == q.astype(object)).all()
    # the extreme the scaling admits: a term rounding up to exactly 2^54
    ds = digits_of(np.array([2 ** BITS], dtype=np.int64))
    assert int(ds[-1][0]) == 64 and all(int(d[0]) == 0 for d in ds[:-1])


def test_accumulators_and_the_recombination_are_exact():
    m = 2048                                                 # items: 2 m indicator columns, at most m of them set per respondent
    assert m * 128 < 2 ** 31                                 # an int32 plane sum, every digit at its extreme
    acc = np.array([m * 127, -m * 128, m * 127, -m * 128, m * 127, -m * 128, m * 64], dtype=np.int64)
    hi = ((acc[6] * 256 + acc[5]) * 256 + acc[4]) * 256 + acc[3]
    lo = (acc[2] * 256 + acc[1]) * 256 + acc[0]
    assert abs(int(hi)) < 2 ** 53 and abs(int(lo)) < 2 ** 53  # both exact as doubles
    assert float(hi) == int(hi) and float(lo) == int(lo)
    exact = sum(int(a) * 256 ** s for s, a in enumerate(acc))
    got = float(hi) * 16777216.0 + float(lo)                 # one rounding
    assert abs(Fraction(got) - exact) <= Fraction(np.spacing(abs(got))) / 2


def test_the_sum_is_within_the_stated_bound_of_the_exact_sum_of_the_fp64_terms():
    rng = np.random.default_rng(2)
    m = 777
    f = rng.standard_normal(m) * 3.0
    gp = np.log(1.0 + np.exp(-f))                            # the fp64 terms (-G+), as stages.hip / theta_fixed.hip form them
    gm = np.log(1.0 + np.exp(f))
    e = row_exponent(np.abs(f).max())
    qp, qm = quantise(gp, e), quantise(gm, e)
    for _ in range(50):
        y = rng.choice([1, -1, 0], size=m, p=[0.45, 0.45, 0.1])
        sel = np.where(y == 1, qp, np.where(y == -1, qm, 0)).astype(object)
        total = int(sel.sum())                               # what the seven planes add up to, exactly
        acc = [int(np.where(y == 1, dp, np.where(y == -1, dm, 0)).sum()) for dp, dm in zip(digits_of(qp), digits_of(qm))]
        assert sum(a * 256 ** s for s, a in enumerate(acc)) == total
        hi = ((acc[6] * 256 + acc[5]) * 256 + acc[4]) * 256 + acc[3]
        lo = (acc[2] * 256 + acc[1]) * 256 + acc[0]
        got = -((float(hi) * 16777216.0 + float(lo)) * 2.0 ** (e - BITS))
        exact = -(sum(Fraction(float(v)) for v in gp[y == 1]) + sum(Fraction(float(v)) for v in gm[y == -1]))
        bound = Fraction(int((y != 0).sum())) * Fraction(2) ** (e - 55) + Fraction(np.spacing(abs(got)))
        assert abs(Fraction(got) - exact) <= bound
        # ... and does not depend on the order of the items
        p = rng.permutation(m)
        assert int(np.where(y[p] == 1, qp[p], np.where(y[p] == -1, qm[p], 0)).astype(object).sum()) == total
