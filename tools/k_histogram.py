"""Distribution of the slice loops' rejection counts k under the default contract (what the predictor's candidate windows have to
cover): histogram of k and of k_j + k_{j+1} over the iterations behind a burn-in.
    python tools/k_histogram.py [n = 8192] [m = 1024] [burn-in = 100] [iterations = 40]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
burn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
its = int(sys.argv[4]) if len(sys.argv) > 4 else 40
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
s = Sampler(h, y, th0, rng="reference", rstream=RStream(20240), theta_stabilise=True, fstar_fused=False)
s.init(); s.check()
for i in range(burn): s.step()
s.check()
h1 = np.zeros(64, dtype=np.int64); h2 = np.zeros(128, dtype=np.int64)
ks = []
for i in range(its):
    s.step(); s.check()
    k = s.get("ess_k").astype(np.int64)
    ks.append(k.copy())
    h1 += np.bincount(k, minlength=64)[:64]
    h2 += np.bincount(k[:-1] + k[1:], minlength=128)[:128]
print("k:      ", {i: int(v) for i, v in enumerate(h1) if v})
print("k + k': ", {i: int(v) for i, v in enumerate(h2) if v})
ks = np.array(ks)
print("mean %.2f sd %.2f; per-item mean k: sd across items %.2f; lag-1 correlation of an item's k across iterations %.3f" % (
    ks.mean(), ks.std(), ks.mean(0).std(), np.corrcoef(ks[:-1].ravel(), ks[1:].ravel())[0, 1]))
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "k_hist_%d_%d.npy" % (n, m)), ks)
