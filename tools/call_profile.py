"""cProfile of one whole gpirtMCMC() call (Python wrapper + the C call): where do the seconds that are not iterations go?
    python tools/call_profile.py [n = 8192] [m = 1024]"""
import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpirt_amd import gpirtMCMC
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
y, th0 = make_responses(n, m, seed=20240)
kw = dict(vote_codes=dict(yea=[1], nay=[-1], missing=[None]), theta_init=th0, rng="item", seed=7, theta_stabilise=True)
gpirtMCMC(y, 1, 0, **kw)
pr = cProfile.Profile(); pr.enable()
gpirtMCMC(y, 1, 0, **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
