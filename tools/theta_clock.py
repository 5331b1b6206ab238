"""In-kernel stamps of the fixed-point log-posterior product (gpirt_debug_theta_clock): the shader clock the chip holds under
the int8 MFMA kernel, the time of a work-group's main loop and of its prologue / epilogue.
    python tools/theta_clock.py [n = 8192] [m = 1024]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpirt_amd.ops import Handle, to_device, check, _p
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = np.random.default_rng(1)
fstar = to_device(np.asfortranarray(rng.standard_normal((1001, m)) * 3.0))
y = np.where(rng.random((n, m)) < 0.5, 1.0, -1.0); y[rng.random((n, m)) < 0.05] = np.nan
yd = to_device(np.asfortranarray(y))
h = Handle()
st = np.zeros(1024 * 6, dtype=np.int64)
for rep in range(30):                      # back-to-back launches: the clock settles
    check(h.lib.gpirt_debug_theta_clock(h._h, _p(yd), _p(fd := fstar), n, m, C.c_void_p(st.ctypes.data), st.size))
wgs = min(1024, 32 * ((n + 255) // 256))
s = st.reshape(1024, 6)[:wgs]
loop_cyc, loop_us = s[:, 2] - s[:, 0], (s[:, 3] - s[:, 1]) / 100.0
tail_us = (s[:, 5] - s[:, 3]) / 100.0
ghz = loop_cyc / (loop_us * 1e3)
t0 = s[:, 1].min()
print(f"n = {n}, m = {m}: {wgs} work-groups stamped")
print(f"main loop: median {np.median(loop_us):.2f} us, {np.median(loop_cyc):.0f} shader cycles -> clock {np.median(ghz):.2f} GHz (min {ghz.min():.2f}, max {ghz.max():.2f})")
chunks = ((2 * ((m + 15) // 16 * 16) + 31) // 32 + 7) // 8 * 2
print(f"   = {np.median(loop_cyc) / chunks:.0f} cycles per chunk of 56 MFMAs (1792 at the MFMA's 32 cycles)")
print(f"epilogue: median {np.median(tail_us):.2f} us; kernel span {((s[:, 5].max() - t0) / 100.0):.1f} us; starts at {np.percentile((s[:, 1] - t0) / 100.0, [0, 25, 50, 75, 100]).round(1).tolist()} us")
