"""Time of the factorisation alone (operator entry: K + jitter + potrf) at n, best and mean of `reps` single runs.
All GPIRT_* switches are read by the library: the target of tools/sweep_env.sh-style sweeps.
    python tools/factor_time.py [n = 8192] [reps = 20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h = Handle()
theta = torch.from_numpy(make_responses(n, 2, seed=1)[1]).cuda()
ignore = os.environ.get("GPIRT_X_IGNORE") == "1"      # timing experiments that break the factor on purpose
def factor():
    try:
        h.factor(theta)
    except RuntimeError:
        if not ignore:
            raise
for _ in range(3):
    factor()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); factor(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("GPIRT_"))
print(f"n={n} factor: min {min(ts):.3f} ms  mean {sum(ts)/len(ts):.3f} ms   [{tag}]", flush=True)
