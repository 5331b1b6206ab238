"""Host-side logic that needs no GPU: data preparation (the reference's own testthat cases for
response_matrix, tests/testthat/test_response_matrix.R:53-100), synthetic data, item partition."""
import warnings

import numpy as np
import pytest

from gpirt_amd.response_matrix import as_response_matrix, is_response_matrix, response_matrix

CODES = dict(yea=[1], nay=[0], missing=[None])


def test_response_matrix_basic():
    x = np.array([[1, 1], [0, 0], [1, np.nan]], dtype=float)
    r = response_matrix(x, CODES)
    assert is_response_matrix(r)
    vals = np.asarray(r)
    assert set(np.unique(vals[~np.isnan(vals)])) == {-1.0, 1.0} and np.isnan(vals).sum() == 1


def test_response_matrix_multiple_yea_codes():
    x = np.array([[1, 3], [-1, -1], [2, np.nan]], dtype=float)
    r = np.asarray(response_matrix(x, dict(yea=[1, 2, 3], nay=[-1], missing=[None])))
    assert np.array_equal(r[:, 0], [1, -1, 1]) and r[0, 1] == 1 and r[1, 1] == -1 and np.isnan(r[2, 1])


def test_response_matrix_strings_and_unknown_codes():
    x = np.array([["Yea", "Yea"], ["Nay", "Nay"], ["Yes", None]], dtype=object)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = np.asarray(response_matrix(x, dict(yea=["Yea"], nay=["Nay"], missing=[None])))
    assert any("treated as missing" in str(v.message) for v in w)
    assert np.isnan(r[2, 0]) and np.isnan(r[2, 1])


def test_response_matrix_drops_unanimous_and_rejects_lists():
    x = np.array([[1, 1, 4], [4, 1, 1], [1, 1, 4]], dtype=float)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = response_matrix(x)
    assert r.shape == (3, 2) and any("unanimous" in str(v.message) for v in w)
    with pytest.raises(TypeError):
        response_matrix([[1, 0], [0, 1]])


def test_as_response_matrix_is_idempotent():
    x = np.array([[1, 0], [0, 1], [1, 1]], dtype=float)
    r = response_matrix(x, CODES)
    assert as_response_matrix(r) is r
    assert np.array_equal(np.asarray(as_response_matrix(x, CODES)), np.asarray(r), equal_nan=True)


def test_synthetic_generator():
    from gpirt_amd.synthetic import make_responses
    y, th = make_responses(200, 30, seed=1)
    assert y.shape == (200, 30) and y.flags.f_contiguous and th.shape == (200,)
    ok = ~np.isnan(y)
    assert set(np.unique(y[ok])) == {-1.0, 1.0} and 0.02 < (~ok).mean() < 0.09
    for j in range(30):
        assert len(np.unique(y[ok[:, j], j])) == 2
    k = (th + 5) / 0.01
    assert np.abs(k - np.rint(k)).max() < 1e-9
    y2, th2 = make_responses(200, 30, seed=1)
    assert np.array_equal(y, y2, equal_nan=True) and np.array_equal(th, th2)


def test_item_range_partitions_everything():
    from gpirt_amd.distributed import item_range
    for m in (1, 7, 418, 1024):
        for w in (1, 2, 3, 8):
            got = [item_range(m, r, w) for r in range(w)]
            assert got[0][0] == 0 and got[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1
