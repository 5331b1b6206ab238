"""The all-core driver of the CPU oracle lives in oracle/parallel.py (bench.py's cpu_baseline uses it too); the tests keep
importing it under this name."""
from oracle.parallel import *          # noqa: F401,F403
from oracle.parallel import host_cores, factor, draw_f, draw_fstar, draw_theta, draw_beta          # noqa: F401
