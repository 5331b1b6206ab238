"""Derive the senate116 response matrix (config C1) from the reference's raw CSVs.

Follows data-raw/senate116.R:5-8 (first-session roll calls), the vignette's reshape
(vignettes/gpirt-vignette.Rmd:134-145: rows = icpsr sorted, columns = rollnumber sorted) and
R/response_matrix.R:79-90 (default codes, unanimous items dropped).
Output: tests/golden/senate116_y.npz  (int8: +1 / -1 / 0 = NA), n=100, m=418.
Run in the build container only (needs /root/reference); the fixture is committed.
"""
import csv
import os
import sys

import numpy as np

REF = "/root/reference/data-raw"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from gpirt_amd.response_matrix import response_matrix  # noqa: E402

with open(os.path.join(REF, "S116_rollcalls.csv")) as fh:
    session1 = {int(r["rollnumber"]) for r in csv.DictReader(fh) if int(r["session"]) == 1}
votes = []
with open(os.path.join(REF, "S116_votes.csv")) as fh:
    for r in csv.DictReader(fh):
        rn = int(r["rollnumber"])
        if rn in session1:
            votes.append((int(r["icpsr"]), rn, int(r["cast_code"])))
icpsr = sorted({v[0] for v in votes})
rolls = sorted({v[1] for v in votes})
raw = np.full((len(icpsr), len(rolls)), np.nan)
ri = {v: i for i, v in enumerate(icpsr)}
ci = {v: i for i, v in enumerate(rolls)}
for s, rn, c in votes:
    raw[ri[s], ci[rn]] = c
import warnings
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    y = np.asarray(response_matrix(raw))
enc = np.where(np.isnan(y), 0, y).astype(np.int8)
out = os.path.join(os.path.dirname(__file__), "senate116_y.npz")
np.savez_compressed(out, y=enc, icpsr=np.array(icpsr, dtype=np.int32))
print(raw.shape, "->", y.shape, "NA frac", np.isnan(y).mean())
