import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n, m = 8192, 1024
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
for tag, kw in (("fast", dict(preset="fast", seed=20240)), ("aswritten", dict(rng="item", seed=20240, theta_stabilise=True, fstar_fused=False, kstar_rank=0)),
                ("fused", dict(rng="item", seed=20240, theta_stabilise=True, fstar_fused=True, kstar_rank=0)), ("fast2", dict(preset="fast", seed=20240))):
    s = Sampler(h, y, th0, **kw)
    s.init(); s.check()
    for _ in range(4):
        s.step()
    s.check()
    print(tag, "mean k", s.get("ess_k").mean(), flush=True)
    s.close()
