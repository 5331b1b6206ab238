"""In-kernel time stamps of ONE pass of the R-stream replay's draw_f (rs3_products_kernel + rs3_slice_kernel, rng_ess.hip):
    python tools/rs_trace.py [n = 8192] [m = 64] [pass = 5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpirt_amd import _lib
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lib = _lib.load()
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
s = Sampler(h, y, th0, rng="reference", rstream=RStream(20240), theta_stabilise=True, fstar_fused=False)
s.init(); s.check(); s.step(); s.check()
_lib.check(lib.gpirt_debug_rs_trace(h._h, p))
s.step(); s.check()
import ctypes as C
t = np.empty(128, dtype=np.int64)
_lib.check(lib.gpirt_sampler_get(s._s, b"rs_trace", C.c_void_p(t.ctypes.data), 128))
_lib.check(lib.gpirt_debug_rs_trace(h._h, -1))
if os.environ.get("GPIRT_RS_PREDICT") in ("2", "3"):
    ns = int(t[63])
    st = t[:ns].astype(float) / 100.0
    print("slice kernel, work-group 0 (us from its start):", np.round(st - st[0], 2).tolist())
    sl = t[32:32 + ns].astype(float) / 100.0
    print("slice kernel, LAST work-group (us from work-group 0's start):", np.round(sl - st[0], 2).tolist())
    for b in range(3):
        q = t[64 + 8 * b: 64 + 8 * b + 6].astype(float) / 100.0
        print(f"products work-group {b} (0 / middle / last full): anchor {q[1]-q[0]:.2f}, windows staged +{q[2]-q[1]:.2f}, MFMAs +{q[3]-q[2]:.2f}, "
              f"barrier +{q[4]-q[3]:.2f}, sum + store +{q[5]-q[4]:.2f}; started {q[0]-t[64]/100.0:.2f} us after work-group 0")
    print("slice kernel started %.2f us after the products' work-group 0" % (st[0] - t[64] / 100.0))
else:
    # the predictor's decide kernel (rs_predict.hip): start, walk + cos / sin done, terms done, ticket taken [, parts read, decided]
    for b, name in ((0, "work-group 0"), (16, "the last work-group of the grid")):
        q = t[b:b + 6].astype(float) / 100.0
        q = q[q > 0]
        if len(q) == 0:
            print(f"decide kernel, {name}: no stamps (the pass asked for found every item predicted?)")
            continue
        print(f"decide kernel, {name} (us from its start):", np.round(q - q[0], 2).tolist())
        w = t[b + 6:b + 10].astype(float) / 100.0
        if (w > 0).all():
            print(f"    its wave 0: rows issued +{w[0] - q[0]:.2f}, uniforms in LDS +{w[1] - w[0]:.2f}, walk +{w[2] - w[1]:.2f}, cos / sin +{w[3] - w[2]:.2f}")
    for b in range(3):
        q = t[64 + 8 * b: 64 + 8 * b + 5].astype(float) / 100.0
        cyc = int(t[64 + 8 * b + 6] - t[64 + 8 * b + 5])
        print(f"predictor products unit {b} (first / middle / last full): windows staged +{q[1]-q[0]:.2f}, MFMAs issued +{q[2]-q[1]:.2f}, barrier +{q[3]-q[2]:.2f}, "
              f"sum + store +{q[4]-q[3]:.2f}; started {q[0]-t[64]/100.0:.2f} us after unit 0; shader clock {cyc / max(q[4]-q[0], 1e-9):.0f} MHz")
    print("decide kernel, the LAST ARRIVER (work-group %d): ticket returned %.2f us after work-group 0 started, parts read +%.2f, decided +%.2f; work-group 255 started %.2f us after work-group 0"
          % (t[36], (t[32] - t[0]) / 100.0, (t[33] - t[32]) / 100.0, (t[34] - t[33]) / 100.0, (t[16] - t[0]) / 100.0))
