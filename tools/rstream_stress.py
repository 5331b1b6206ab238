"""Stress of the replay's speculative draw_f against the oracle: small odd shapes (one item, two items, n not a multiple of
32, the redo path at the first / last item) with the candidate limit lowered so that candidates run out all the time.
    python tools/rstream_stress.py"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpirt_amd import Sampler, _lib
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.synthetic import make_responses
from oracle import oracle
oracle.build()
lib = _lib.load()
h = Handle()
bad = 0
for n, m, limit in itertools.product((64, 66, 100, 130, 258, 1000, 1026, 2050), (1, 2, 3, 7), (0, 1, 2)):
    y, th0 = make_responses(n, m, seed=1000 * n + m)
    _lib.check(lib.gpirt_debug_rs_cand_limit(h._h, limit))
    rs = RStream(n + m + limit)
    s = Sampler(h, y, th0, rng="reference", rstream=rs)
    s.init()
    for _ in range(3):
        s.step()
    s.check()
    got = {k: s.get(k) for k in ("theta", "f", "beta")}
    ks = s.get("ess_k"); st = rs.state(); s.close()
    r = oracle.RStream(n + m + limit)
    ref = oracle.gpirt_mcmc(r, y, th0, 3, 0)
    mt_ref, mti_ref = r.mt_state()
    ok = (np.array_equal(got["theta"], ref["theta"][3]) and np.abs(got["f"] - ref["f"][:, :, 3]).max() <= 1e-9 and
          np.abs(got["beta"] - ref["beta"][:, :, 3]).max() <= 1e-9 and st[1] == mti_ref and np.array_equal(st[0], mt_ref))
    bad += not ok
    print(f"n {n:5d} m {m} limit {limit}: {'ok ' if ok else 'MISMATCH'}  k = {ks.tolist()}", flush=True)
_lib.check(lib.gpirt_debug_rs_cand_limit(h._h, 0))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
