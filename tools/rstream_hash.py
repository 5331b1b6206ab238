"""sha256 of the chain state after a few iterations of the DEFAULT contract (R-stream replay), for several shapes and both forms
of its draw_f (GPIRT_RS_PREDICT=1: predict + verify, 2: every pass in fp64): the target of bit-identity comparisons between
builds of the library (tests/test_gpu_fences.py: the fenced reference forms of the meetings).
    python tools/rstream_hash.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
h = Handle()
for n, m, its in ((300, 14, 3), (1030, 9, 2), (2048, 48, 2), (8192, 24, 2)):
    y, th0 = make_responses(n, m, seed=n + m)
    for mode in (1, 2):
        with h.config("GPIRT_RS_PREDICT", mode):
            rs = RStream(77)
            s = Sampler(h, y, th0, rng="reference", rstream=rs, theta_stabilise=True)
            s.init()
            for _ in range(its):
                s.step()
            s.check()
            hh = hashlib.sha256()
            for name in ("f", "theta", "beta", "ess_k"):
                hh.update(np.ascontiguousarray(s.get(name)).tobytes())
            st = rs.state()
            hh.update(np.ascontiguousarray(st[0]).tobytes()); hh.update(str(st[1]).encode())
            print(f"replay n={n} m={m} predict={mode}: {hh.hexdigest()}", flush=True)
            s.close()
