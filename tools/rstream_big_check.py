"""The replay's draw_f above n = 8192 (the slice loop's rows no longer fit its work-groups' registers: memory path) against
the oracle, one iteration.   python tools/rstream_big_check.py [n = 9216] [m = 2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpirt_amd import Sampler
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.synthetic import make_responses
from oracle import oracle
oracle.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2
y, th0 = make_responses(n, m, seed=3)
h = Handle(); rs = RStream(31)
s = Sampler(h, y, th0, rng="reference", rstream=rs, theta_stabilise=True)
s.init(); s.step(); s.check()
got = {k: s.get(k) for k in ("theta", "f", "beta")}; st = rs.state(); s.close()
t0 = time.time()
r = oracle.RStream(31)
ref = oracle.gpirt_mcmc(r, y, th0, 1, 0, theta_stabilise=True, blocked_potrf=True, nthreads=16)
print(f"oracle {time.time() - t0:.0f} s")
mt_ref, mti_ref = r.mt_state()
print("theta equal", np.array_equal(got["theta"], ref["theta"][1]), " max|df|", np.abs(got["f"] - ref["f"][:, :, 1]).max(),
      " max|dbeta|", np.abs(got["beta"] - ref["beta"][:, :, 1]).max(), " stream state equal", st[1] == mti_ref and np.array_equal(st[0], mt_ref))
