"""R-stream replay (the default contract) at growing sizes: rate, and on failure what the state looks like.
    python tools/rstream_probe.py [n m] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(2048, 256), (4096, 512), (8192, 256), (8192, 1024)]
h = Handle()
for n, m in sizes:
    y, th0 = make_responses(n, m, seed=20240)
    s = Sampler(h, y, th0, rng="reference", rstream=RStream(20240), theta_stabilise=False, fstar_fused=False)
    s.init(); s.check()
    print(f'{n} x {m}: after init nonfinite f {np.sum(~np.isfinite(s.get("f")))} fstar {np.sum(~np.isfinite(s.get("fstar")))} L {np.sum(~np.isfinite(s.get("L")))}', flush=True)
    try:
        s.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            s.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        s.check()
        print(f"{n} x {m}: {dt * 1e3:.1f} ms per iteration ({1 / dt:.2f} it/s), mean k {s.get('ess_k').mean():.2f}", flush=True)
    except Exception as e:
        k = s.get("ess_k"); nu = s.get("nu"); f = s.get("f"); th = s.get("theta"); fs = s.get("fstar")
        print(f"{n} x {m}: FAILED {e}\n   ess_k max {k.max()} argmax {k.argmax()} nonfinite: nu[:,0] {np.sum(~np.isfinite(nu[:, 0]))} f {np.sum(~np.isfinite(f))} "
              f"(first bad column {np.where(~np.isfinite(f).all(axis=0))[0][:3]}) theta {np.sum(~np.isfinite(th))} fstar {np.sum(~np.isfinite(fs))}", flush=True)
    s.close()
