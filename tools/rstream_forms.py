import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n, m = 8192, 1024
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
for kw in (dict(fstar_fused=False), dict(fstar_fused=True), dict(fstar_fused=True, kstar_rank=64)):
    try:
        s = Sampler(h, y, th0, rng="reference", rstream=RStream(20240), theta_stabilise=True, **kw)
        s.init(); s.check(); s.step(); s.check()
        s.enable_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): s.step()
        s.check(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(kw, f"{dt*1e3:.2f} ms/iter", {k: round(v, 2) for k, v in s.stage_times().items()}, flush=True)
        s.close()
    except Exception as e:
        print(kw, "FAILED", repr(e)[:300], flush=True)
