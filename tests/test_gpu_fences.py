"""The panel kernel's fence-free hand-off (every handed-off byte stored and loaded sc1, flagsync.h) against the fenced
reference form (agent-scope release before each counter, acquire after each wait: -DGPIRT_PANEL_FENCES, built as a
second library by `make -C gpirt_amd/csrc fences`): the factor must be BIT-IDENTICAL -- the arithmetic is the same, only
the way bytes travel between work-groups differs.  Sizes: ragged single panel (257), several panels with a ragged end
(1000, 2500), the metric size (8192), and the bordered factorisations of the sampler (64 rows at n = 8192 and 16384,
1024 rows at n = 4096).  Round-2 advisor finding: the fenced form was never built or tested."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hashes(lib=None, **switches):
    env = dict(os.environ)
    env.pop("GPIRT_HIP_LIBRARY", None)
    if lib:
        env["GPIRT_HIP_LIBRARY"] = lib
    env.update(switches)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "factor_hash.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith(("operator", "sampler"))]


_DEFAULT = []


def _default_hashes():
    if not _DEFAULT:
        _DEFAULT.append(_hashes())
    return _DEFAULT[0]


def test_fenced_build_gives_the_same_factor_bit_for_bit():
    from gpirt_amd import build
    lib = build.build_fences()
    assert os.path.exists(lib)
    a = _default_hashes()
    b = _hashes(lib)
    assert len(a) == len(b) == 7
    assert a == b, "\n".join(f"{x}\n{y}" for x, y in zip(a, b) if x != y)


@pytest.mark.parametrize("switch", ["GPIRT_DEFER=2", "GPIRT_LOOKAHEAD=2"])
def test_schedule_switches_leave_the_factor_bit_identical(switch):
    """The schedules of the factorisation that apply the SAME products in the same order per element (potrf.hip: the plain
    right-looking order instead of the deferred updates; no look-ahead side stream) must reproduce the default factor bit
    for bit -- operator sizes 257 ... 8192 and the sampler's bordered factorisations.  (The environment is read once per
    process: each switch runs in a child.)"""
    k, v = switch.split("=")
    a = _default_hashes()
    b = _hashes(**{k: v})
    assert len(a) == len(b) == 7
    assert a == b, "\n".join(f"{x}\n{y}" for x, y in zip(a, b) if x != y)


def test_fenced_build_replays_r_stream_draws_bit_for_bit():
    """The meetings of the R-stream replay's draw_f -- the slice kernel's flag meeting (rng_ess.hip) and the predictor's ticket
    (rs_predict.hip) -- hand partial sums between work-groups with write-through stores, a wait, and sc1 loads.  The second
    library holds their fenced reference forms (agent-scope release in front of the flag / ticket, acquire behind the poll /
    in the last arriver): every replayed draw -- f, theta, beta, rejection counts, the generator's position -- must be the
    default build's bit for bit, with the predictor (GPIRT_RS_PREDICT=1) and without (2), at four shapes."""
    from gpirt_amd import build
    lib = build.build_fences()

    def run(libpath):
        env = dict(os.environ)
        env.pop("GPIRT_HIP_LIBRARY", None)
        if libpath:
            env["GPIRT_HIP_LIBRARY"] = libpath
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rstream_hash.py")], env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return [ln for ln in r.stdout.splitlines() if ln.startswith("replay")]
    a, b = run(None), run(lib)
    assert len(a) == len(b) == 8
    assert a == b, "\n".join(f"{x}\n{y}" for x, y in zip(a, b) if x != y)
