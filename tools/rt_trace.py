"""In-kernel stamps of ONE sub-panel kernel inside the dependency-driven factorisation (GPIRT_RUNTIME=2): when each row
block of the launch started and ended -- do its work-groups find their compute units at once?
    python tools/rt_trace.py [n = 8192] [k0 = 0]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpirt_amd.ops import Handle
from gpirt_amd._lib import check
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
y, th0 = make_responses(n, 8, seed=n)
th0 = -5.0 + np.clip(np.rint((th0 + 5.0) / 0.01), 0, 1000) * 0.01
h = Handle()
h.config_set("GPIRT_RUNTIME", int(os.environ.get("RT", "2")))
s = Sampler(h, y, th0, rng="item", seed=1, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
s.init(); s.check()
for _ in range(2):
    s.factor()
s.check()
nrb = 200
count = nrb * 40 * 8
check(h.lib.gpirt_debug_panel_trace(h._h, k0, None, count))
check(h.lib.gpirt_debug_rt_census(h._h, None))
s.factor(); s.check()
raw = np.zeros(2 * 4096 * 4 * 4 + 4096 * 8 * 8 + 65536 * 16, dtype=np.uint8)
check(h.lib.gpirt_debug_rt_census(h._h, C.c_void_p(raw.ctypes.data)))
cen = raw[: 2 * 4096 * 16].view(np.uint32).reshape(2, 4096, 4)
st = raw[2 * 4096 * 16: 2 * 4096 * 16 + 4096 * 64].view(np.int64).reshape(4096, 8)
tt = raw[2 * 4096 * 16 + 4096 * 64:].view(np.int64).reshape(65536, 2)
ok = (tt[:, 0] > 0) & (tt[:, 1] > 0) & (tt[:, 0] < 2**62)
if ok.any():
    t0t = tt[ok, 0].min()
    idx = np.where(ok)[0]
    print(f"task stamps: {len(idx)} tasks; first start 0, last end {(tt[ok, 1].max() - t0t) / 100.0:.0f} us")
    # running tasks over time (50 us bins) and the first / last start of consecutive blocks of 500 tasks in list order
    T = (tt[ok, 1].max() - t0t) / 100.0
    bins = np.arange(0, T + 100, 100.0)
    busy = np.zeros(len(bins))
    for a_, b_ in zip((tt[ok, 0] - t0t) / 100.0, (tt[ok, 1] - t0t) / 100.0):
        i0, i1 = int(a_ // 100), int(b_ // 100)
        for i in range(i0, i1 + 1):
            lo, hi = max(a_, bins[i]), min(b_, bins[i] + 100.0)
            if hi > lo: busy[i] += (hi - lo) / 100.0
    print("  workers busy per 100 us:", " ".join(f"{int(round(x))}" for x in busy[:80]))
    for lo in range(0, int(idx.max()) + 1, 1000):
        sel = ok.copy(); sel[:lo] = False; sel[lo + 1000:] = False
        if sel.any():
            print(f"  tasks {lo:6d}..: start {((tt[sel, 0].min() - t0t) / 100.0):8.0f} .. {((tt[sel, 0].max() - t0t) / 100.0):8.0f}  mean dur {((tt[sel, 1] - tt[sel, 0]).mean() / 100.0):6.1f} us")
st = st[(st[:, 3] > 0) & (st[:, 3] < 10**12)]
if len(st):
    us = lambda x: x / 100.0
    print(f"worker stats ({len(st)} workers): tasks per worker mean {st[:,0].mean():.1f} (urgent {st[:,4].mean():.1f}); per worker: in tasks {us(st[:,1].mean()):.0f} us, "
          f"dequeue + idle {us(st[:,2].mean()):.0f} us, alive {us(st[:,3].mean()):.0f} us; mean task {us(st[:,1].sum() / max(1, st[:,0].sum())):.1f} us")
for kind, name in ((0, "workers"), (1, "holders")):
    c = cen[kind]; c = c[c[:, 3] != 0xffffffff]
    xcc = c[:, 1] & 0xf
    cu = (c[:, 0] >> 8) & 0xf; se = (c[:, 0] >> 13) & 0x7; sh = (c[:, 0] >> 12) & 1
    key = xcc.astype(np.int64) * 1000 + se * 100 + sh * 50 + cu
    stay = c[:, 3] == 1
    print(f"{name}: {len(c)} recorded, stayed {int(stay.sum())}; per XCC stayed {[int(((xcc == x) & stay).sum()) for x in range(8)]}, "
          f"left {[int(((xcc == x) & ~stay).sum()) for x in range(8)]}; distinct CUs of the stayers per XCC {[len(set(key[(xcc == x) & stay])) for x in range(8)]}")
    if kind == 0:
        hk = cen[1]; hk = hk[hk[:, 3] != 0xffffffff]
        hkey = set(((hk[:, 1] & 0xf).astype(np.int64) * 1000 + ((hk[:, 0] >> 13) & 7) * 100 + ((hk[:, 0] >> 12) & 1) * 50 + ((hk[:, 0] >> 8) & 0xf)).tolist())
        on_res = [k in hkey for k in key.tolist()]
        print(f"   workers that stayed on a holder's CU: {int((np.array(on_res) & stay).sum())}; that left from one: {int((np.array(on_res) & ~stay).sum())}")
tr = np.zeros(count, dtype=np.int64)
check(h.lib.gpirt_debug_panel_trace(h._h, k0, C.c_void_p(tr.ctypes.data), count))
tr = tr.reshape(nrb, 40, 8)
first = np.where(tr[:, :, 0] > 0, tr[:, :, 0], np.iinfo(np.int64).max).min(axis=1)
last = tr.max(axis=(1, 2))
live = last > 0
t0 = first[live].min()
print(f"sub-panel at column {k0}: {int(live.sum())} row blocks stamped; us since the first stamp")
for R in np.where(live)[0]:
    print(f"  R{int(R):3d} start {(first[R] - t0) / 100.0:8.1f}  end {(last[R] - t0) / 100.0:8.1f}")
check(h.lib.gpirt_debug_panel_trace(h._h, -1, None, 0))
