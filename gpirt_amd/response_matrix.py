"""Host-side mirror of the reference's `response_matrix` S3 helpers (R/response_matrix.R:51-127).

Pure data preparation that runs once, upstream of the drop-in boundary: recode raw responses to
{+1, -1, NaN} doubles and drop unanimous items.  In an R deployment this stays in R unchanged
(INTEGRATION.md); the Python mirror exists so `gpirtMCMC()` here keeps the reference's signature.
"""
from __future__ import annotations

import warnings

import numpy as np

DEFAULT_CODES = dict(yea=(1, 2, 3), nay=(4, 5, 6), missing=(0, 7, 8, 9, None))


class ResponseMatrix(np.ndarray):
    """float64 (n x m) array holding only +1 / -1 / NaN (the reference's class tag)."""


def _isin(a, codes):
    codes = [c for c in np.atleast_1d(np.array(list(codes), dtype=object)) if c is not None]
    out = np.zeros(a.shape, dtype=bool)
    for c in codes:
        if isinstance(c, float) and np.isnan(c):
            continue
        out |= (a == c)
    return out


def response_matrix(data, response_codes=None):
    """R/response_matrix.R:51-99: yea -> +1, nay -> -1, missing/unknown -> NaN, drop unanimous."""
    if isinstance(data, (list, tuple, dict)):
        raise TypeError("Conversion from lists to response_matrix objects is currently unsupported.")
    codes = dict(DEFAULT_CODES if response_codes is None else response_codes)
    raw = np.asarray(data)
    if raw.ndim != 2:
        raise ValueError("data must be a 2-d array (respondents x items)")
    if raw.dtype.kind in "fiub":                       # numeric input: vectorised recode
        num = np.asarray(raw, dtype=np.float64)        # (no copy when the data are doubles already: nothing below writes to it)
        isna = np.isnan(num)

        def _num_codes(cs):
            out = []
            for c in np.atleast_1d(np.array(list(cs), dtype=object)):
                if c is None or (isinstance(c, float) and np.isnan(c)):
                    continue
                try:
                    out.append(float(c))
                except (TypeError, ValueError):
                    pass                               # a string code can never match a number
            return np.array(out, dtype=np.float64)

        def _any_of(cs):                               # one comparison pass per code, in the array's own memory order
            out = np.zeros_like(num, dtype=bool)       # (same memory order as num)
            for c in _num_codes(cs):                   # (np.isin flattens in C order: an 8192 x 1024 column-major matrix was
                out |= (num == c)                      # copied six times, 1.2 s of a 1.8 s call)
            return out

        yea = _any_of(codes["yea"]) & ~isna
        nay = _any_of(codes["nay"]) & ~isna
        mis = _any_of(codes["missing"]) | isna
        obj = num
    else:
        obj = raw.astype(object)
        isna = np.array([[v is None or (isinstance(v, float) and np.isnan(v)) for v in row] for row in obj],
                        dtype=bool).reshape(obj.shape)
        yea = _isin(obj, np.atleast_1d(codes["yea"])) & ~isna
        nay = _isin(obj, np.atleast_1d(codes["nay"])) & ~isna
        mis = _isin(obj, np.atleast_1d(codes["missing"])) | isna
    unknown = ~(yea | nay | mis)
    if unknown.any():                                  # :72-77
        vals = sorted({str(v) for v in obj[unknown]})
        warnings.warn("Responses with value " + ", ".join(vals) + " were not given a response "
                      "code and will be treated as missing.")
    # :79-81 -- yea -> 1, nay -> -1, missing -> NA, applied in that order (a code listed twice ends up as the later one)
    # (built in the masks' own memory order: np.where on column-major masks writes a row-major result through strides)
    res = np.full_like(yea, np.nan, dtype=np.float64)
    np.copyto(res, 1.0, where=yea)
    np.copyto(res, -1.0, where=nay)
    np.copyto(res, np.nan, where=mis)
    has_pos = np.any(res == 1.0, axis=0)               # :87-90  length(unique(na.omit(x))) == 1
    has_neg = np.any(res == -1.0, axis=0)
    keep = ~(has_pos ^ has_neg)
    if (~keep).any():
        idx = ", ".join(str(i + 1) for i in np.nonzero(~keep)[0])
        warnings.warn(f"Item(s) {idx} discarded as unanimous.")
    if keep.all():
        return np.asfortranarray(res).view(ResponseMatrix)
    return np.asfortranarray(res[:, keep]).view(ResponseMatrix)


def is_response_matrix(x) -> bool:
    """R/response_matrix.R:109-115"""
    if not isinstance(x, ResponseMatrix) or x.ndim != 2:
        return False
    a = np.asarray(x)
    return bool(np.all(np.isnan(a) | (a == 1.0) | (a == -1.0)))


def as_response_matrix(x, response_codes=None):
    """R/response_matrix.R:119-127"""
    if not is_response_matrix(x):
        x = response_matrix(x, response_codes)
    return x
