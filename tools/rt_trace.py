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
# ---- the path behind one outer panel: when were the tile tasks of each kind, by row group, started / finished?
P_SHOW = int(os.environ.get("PANEL", "2"))
tdt = np.dtype([("bi", "<u2"), ("bj", "<u2"), ("kb", "<u2"), ("klen", "<u2"), ("dep0", "<i4"), ("dep1", "<i4"), ("need0", "<u4"), ("need1", "<u4"),
                ("tile", "<i4"), ("prog", "<u2"), ("last", "u1"), ("type", "u1"), ("out", "<i4")])
tasks = np.zeros(65536, dtype=tdt)
n0, n1 = C.c_int(), C.c_int()
check(h.lib.gpirt_debug_rt_tasks(h._h, C.c_void_p(tasks.ctypes.data), 65536, C.byref(n0), C.byref(n1)))
nt = min(65536, n0.value + n1.value)
ok2 = (tt[:nt, 0] > 0) & (tt[:nt, 1] > 0) & (tt[:nt, 0] < 2**62)
if ok2.any():
    T0 = tt[:nt][ok2, 0].min()
    tk = tasks[:nt]
    gi = tk["bi"] // 16
    queue = np.where(np.arange(nt) < n0.value, 0, 1)
    def show(name, sel):
        sel = sel & ok2
        if sel.any():
            print(f"  {name:34s} n {int(sel.sum()):5d}  queue {sorted(set(queue[sel].tolist()))}  start {((tt[:nt][sel, 0].min() - T0) / 100.0):8.0f} .. end {((tt[:nt][sel, 1].max() - T0) / 100.0):8.0f}")
    p = P_SHOW
    print(f"tasks around outer panel {p} (us since the first task started):")
    for g in range(p, p + 5):
        show(f"X = A W^T, A({p}), rows g{g}", (tk["type"] == 1) & (tk["kb"] == 16 * p) & (gi == g))
        show(f"a({p}): K = A({p}), rows g{g}", (tk["type"] == 0) & (tk["kb"] == 16 * p) & (tk["klen"] == 8) & (tk["bj"] // 16 == p) & (gi == g))
        show(f"b1({p}): K = A({p}) -> col {p + 1}, rows g{g}", (tk["type"] == 0) & (tk["kb"] == 16 * p) & (tk["klen"] == 8) & (tk["bj"] // 16 == p + 1) & (gi == g))
        show(f"X = A W^T, B({p}), rows g{g}", (tk["type"] == 1) & (tk["kb"] == 16 * p + 8) & (gi == g))
        show(f"b2({p}): K = B({p}) -> col {p + 1}, rows g{g}", (tk["type"] == 0) & (tk["kb"] == 16 * p + 8) & (tk["klen"] == 8) & (gi == g))
        show(f"c({p}): K = panel {p} -> col {p + 1}, rows g{g}", (tk["type"] == 0) & (tk["kb"] == 16 * p) & (tk["klen"] == 16) & (tk["bj"] // 16 == p + 1) & (gi == g))
        show(f"D({p}) -> col {p + 2}, rows g{g}", (tk["type"] == 0) & (tk["kb"] == 16 * p) & (tk["klen"] == 16) & (tk["bj"] // 16 == p + 2) & (gi == g))
    # ---- the path of ONE row group through every outer panel (ROWS=g): the chain of stages that feeds the sub-panel kernels
    G = int(os.environ.get("ROWS", "-1"))
    if G >= 0:
        print(f"row group g{G} through the panels (us since the first task started):")
        for p in range(0, G + 1):
            show(f"X = A W^T, A({p})", (tk["type"] == 1) & (tk["kb"] == 16 * p) & (gi == G))
            show(f"a({p})", (tk["type"] == 0) & (tk["kb"] == 16 * p) & (tk["klen"] == 8) & (tk["bj"] // 16 == p) & (gi == G))
            show(f"X = A W^T, B({p})", (tk["type"] == 1) & (tk["kb"] == 16 * p + 8) & (gi == G))
            for c in range(p + 1, G + 1):
                show(f"   panel {p} -> col {c}: K = 512 halves", (tk["type"] == 0) & (tk["kb"] // 16 == p) & (tk["klen"] == 8) & (tk["bj"] // 16 == c) & (gi == G))
                show(f"   panel {p} -> col {c}: K = 1024", (tk["type"] == 0) & (tk["kb"] == 16 * p) & (tk["klen"] == 16) & (tk["bj"] // 16 == c) & (gi == G))
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
