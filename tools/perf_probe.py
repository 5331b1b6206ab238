"""Ad-hoc performance probe: stage timings of the sampler and stand-alone operator rates."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpirt_amd.ops import Handle, colmajor, to_device
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses

def ev_time(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

h = Handle()
print("mfma f64 peak TF:", h.calibrate_mfma_f64())
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192]
for n in sizes:
    m = 1024
    theta = torch.from_numpy(make_responses(n, 2, seed=1)[1]).cuda()
    L = h.factor(theta)
    t = ev_time(lambda: h.factor(theta))
    print(f"n={n} factor(K+chol, operator): {t:.3f} ms  -> {n**3/3/t/1e9:.2f} TF")
    S = h.se_kernel(theta, theta, 0.001)
    t = ev_time(lambda: h.se_kernel(theta, theta, 0.001))
    print(f"  se_kernel full: {t:.3f} ms -> {8*n*n/t/1e6:.1f} GB/s")
    Z = colmajor(n, m); Z.normal_()
    t = ev_time(lambda: h.trmm_lz(L, Z))
    print(f"  trmm L*Z m={m}: {t:.3f} ms -> {n*n*m/t/1e9:.2f} TF")
    A = colmajor(n, n); A.normal_()
    t = ev_time(lambda: h.gemm(A, Z))
    print(f"  gemm NN {n}x{m}x{n}: {t:.3f} ms -> {2*n*n*m/t/1e9:.2f} TF")
    t = ev_time(lambda: h.gemm(A, A, tb=True))
    print(f"  gemm NT {n}^3: {t:.3f} ms -> {2*n**3/t/1e9:.2f} TF")
    B = colmajor(n, m + 1001); B.normal_()
    t = ev_time(lambda: h.trsm_lower(L, B))
    print(f"  trsm fwd nrhs={m+1001}: {t:.3f} ms -> {n*n*(m+1001)/t/1e9:.2f} TF")
    t = ev_time(lambda: h.trsm_lower(L, Z, trans=True))
    print(f"  trsm bwd nrhs={m}: {t:.3f} ms -> {n*n*m/t/1e9:.2f} TF")
    del S, A, B
    y, th0 = make_responses(n, m, seed=20240)
    s = Sampler(h, y, th0, rng="item", seed=1)
    s.init(); s.enable_timing(True)
    for _ in range(2): s.step()
    s.check()
    print("  stages(ms):", {k: round(v, 3) for k, v in s.stage_times().items()})
    s.enable_timing(False)
    torch.cuda.synchronize(); t0 = time.time()
    K = 5
    for _ in range(K): s.step()
    s.check(); torch.cuda.synchronize()
    dt = (time.time() - t0) / K
    print(f"  step: {dt*1e3:.2f} ms -> {1/dt:.2f} it/s ; mean ESS rejections {s.get('ess_k').mean():.2f}")
    s.close()
