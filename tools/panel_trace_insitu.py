"""In-kernel stamps of ONE sub-panel kernel as it runs INSIDE the factorisation (beside the updates), next to the same
sub-panel factored with nothing beside it (GPIRT_LOOKAHEAD=2 in a second process is the cleaner comparison; here: the
launch's own stamps).    python tools/panel_trace_insitu.py [n = 8192] [k0 = 2048] [reps = 3]
Prints, relative to the start of diagonal owner 0: when each diagonal owner's potf2 began / was published, and for a few
row blocks when their sweep started, when each step saw L_jj and when it was done -- the in-situ version of
tools/micro/panel_bench.hip's table."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpirt_amd.ops import Handle
from gpirt_amd._lib import check
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
h = Handle()
theta = torch.from_numpy(make_responses(n, 2, seed=1)[1]).cuda()
for _ in range(3):
    h.factor(theta)
nrb = (n - k0 + 63) // 64
count = nrb * 40 * 8
for rep in range(reps):
    check(h.lib.gpirt_debug_panel_trace(h._h, k0, None, count))
    h.factor(theta)
    tr = np.zeros(count, dtype=np.int64)
    check(h.lib.gpirt_debug_panel_trace(h._h, k0, C.c_void_p(tr.ctypes.data), count))
    tr = tr.reshape(nrb, 40, 8)
    t0 = tr[0, 39, 0]
    us = lambda t: (t - t0) / 100.0 if t else float("nan")
    ncb = 8
    print(f"rep {rep}: sub-panel at column {k0}, {nrb} row blocks; times in us since owner 0's potf2 began")
    print("  owners' potf2 [began, published]: " + "  ".join(f"R{R} {us(tr[R,39,0]):.1f}/{us(tr[R,39,1]):.1f}" for R in range(ncb)))
    starts = np.array([us(tr[R, 0, 0]) for R in range(nrb)])
    ends = np.array([us(tr[R, min(R, ncb) - 1, 3]) if R > 0 else us(tr[0, 39, 1]) for R in range(nrb)])
    print(f"  sweep start of row blocks: first {np.nanmin(starts):.1f}, median {np.nanmedian(starts):.1f}, last {np.nanmax(starts):.1f}; "
          f"sweep end: median {np.nanmedian(ends):.1f}, last {np.nanmax(ends):.1f}")
    late = np.argsort(-starts)[:5]
    print("  latest starters: " + ", ".join(f"R{int(r)} at {starts[r]:.1f} (done {ends[r]:.1f})" for r in late))
    for R in (ncb, nrb // 2, nrb - 1):
        print(f"  row block {R}: " + "  ".join(f"j{j} {us(tr[R,j,0]):.0f}>{us(tr[R,j,2]):.0f}>{us(tr[R,j,3]):.0f}" for j in range(ncb)))
check(h.lib.gpirt_debug_panel_trace(h._h, -1, None, 0))
