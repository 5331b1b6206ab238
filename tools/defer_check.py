"""L from the deferred trailing-update schedule (default) must be bit-identical to the undeferred one (GPIRT_DEFER=2).
Runs itself twice (the switch is read once per process) and compares checksums and a residual."""
import os, subprocess, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from gpirt_amd.ops import Handle
    n = int(sys.argv[2])
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    theta = torch.randn(n, generator=g, dtype=torch.float64).cuda()
    h = Handle()
    L = h.factor(theta)
    torch.cuda.synchronize()
    Lc = torch.tril(L).cpu().numpy()
    print(hashlib.sha256(Lc.tobytes()).hexdigest(), float(abs(Lc).sum()))
    sys.exit(0)
for n in (8192, 5000, 3072):
    out = []
    for mode in ("1", "2"):
        env = dict(os.environ, GPIRT_DEFER=mode)
        out.append(subprocess.run([sys.executable, __file__, "child", str(n)], env=env, capture_output=True, text=True).stdout.strip())
    print(n, "identical" if out[0] == out[1] and out[0] else "DIFFERENT", out)
