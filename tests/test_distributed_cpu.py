"""world_size-2 gloo test of the item-sharding host logic (gpirt_amd/distributed.py) on CPU.

Two processes each own half of the item columns of one problem; the sharded run must reproduce the
single-process run draw for draw (theta exactly: the all-reduced log-posterior differs from the
single-process sum only by fp64 re-association), for both Cholesky hand-off modes."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(rank, world, port, chol, outdir, theta="gather", m=7, sub=8):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _oracle_engine import OracleEngine
    OracleEngine.subpanel_width = sub
    from gpirt_amd.distributed import ShardedSampler
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(40, m, seed=4)
    ss = ShardedSampler(OracleEngine, y, th0, dist=dist, chol=chol, theta=theta)
    ss.init()
    for _ in range(2):
        ss.step()
    f = ss.gather("f")
    beta = ss.gather("beta")
    fstar = ss.gather("fstar")
    if rank == 0:
        np.savez(os.path.join(outdir, f"sharded_{chol}_{theta}_{m}_{sub}.npz"), f=f, beta=beta, fstar=fstar, theta=ss.engine.theta,
                 L=ss.engine.L)
    dist.destroy_process_group()


@pytest.mark.parametrize("chol,theta,m,sub", [("replicated", "gather", 7, 8), ("bcast", "gather", 8, 8), ("replicated", "allreduce", 7, 8),
                                              ("bcast", "allreduce", 7, 8), ("distributed", "gather", 8, 8),
                                              ("distributed", "gather", 8, 4)])
def test_two_ranks_reproduce_single_process(tmp_path, chol, theta, m, sub):
    """theta="gather": f* is all-gathered (m = 8: equal shards, flat all-gather; m = 7: unequal shards, the
    all-reduce-of-disjoint-supports fallback) and each rank draws theta for its block of respondents;
    theta="allreduce": the partial log-posteriors are all-reduced.  sub = 4: a sub-panel narrower than half an outer
    panel (16), so the second half of a panel ("the rest", 12 columns) is WIDER than the first -- the broadcast buffers
    must be sized for it (round-3 advisor finding: they held rows x sub doubles and the slice truncated silently)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle_engine import OracleEngine
    from gpirt_amd.distributed import ShardedSampler
    from gpirt_amd.synthetic import make_responses
    port = 29500 + (os.getpid() % 2000) + (1 if chol == "bcast" else 0) + (2 if theta == "gather" else 0)
    mp.spawn(_run, args=(2, port + sub, chol, str(tmp_path), theta, m, sub), nprocs=2, join=True)
    got = np.load(tmp_path / f"sharded_{chol}_{theta}_{m}_{sub}.npz")
    y, th0 = make_responses(40, m, seed=4)
    OracleEngine.subpanel_width = sub
    ref = ShardedSampler(OracleEngine, y, th0, dist=None)
    ref.init()
    for _ in range(2):
        ref.step()
    assert np.array_equal(got["theta"], ref.engine.theta)
    if chol == "distributed":
        # n = 40 -> three 16-column outer panels dealt round-robin to the two ranks, one panel of look-ahead; the
        # NumPy pieces round differently from the oracle's unblocked potrf (the HIP pieces ARE the single-GPU launches:
        # tests/test_gpu_distributed.py asserts bit-identity there)
        assert np.abs(got["L"] - ref.engine.L).max() < 1e-12
        assert np.abs(np.triu(got["L"], 1)).max() == 0
    else:
        assert np.abs(got["L"] - ref.engine.L).max() == 0
    tol = 1e-9 if chol == "distributed" else 1e-12
    assert np.abs(got["f"] - ref.gather("f")).max() < tol
    assert np.abs(got["beta"] - ref.gather("beta")).max() < tol
    assert np.abs(got["fstar"] - ref.gather("fstar")).max() < tol


def _run3(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GPIRT_DIST_DEBUG"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _oracle_engine import OracleEngine
    from gpirt_amd.distributed import ShardedSampler
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(75, 10, seed=14)           # five 16-column outer panels, the last ragged (11 columns = sub-panels of 8 + 3)
    ss = ShardedSampler(OracleEngine, y, th0, dist=dist, chol="distributed")
    ss.init()
    ss.step()
    f, fstar = ss.gather("f"), ss.gather("fstar")
    if rank == 0:
        np.savez(os.path.join(outdir, "sharded3.npz"), f=f, fstar=fstar, theta=ss.engine.theta, L=ss.engine.L)
    dist.destroy_process_group()


def test_three_ranks_pipelined_distributed_factorisation(tmp_path):
    """The factorisation pipelined by halves of a panel (gpirt_amd/distributed.py) with three ranks, five outer panels
    (so ranks own 2 / 2 / 1 of them), a ragged last panel, unequal item shards (10 items over 3 ranks = 3 / 3 / 4: the
    padded tensor-collective gather and the all-reduce form of the f* gather), with the joined-streams debug check on,
    against the single-process run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle_engine import OracleEngine
    from gpirt_amd.distributed import ShardedSampler
    from gpirt_amd.synthetic import make_responses
    port = 29300 + (os.getpid() % 2000)
    mp.spawn(_run3, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    got = np.load(tmp_path / "sharded3.npz")
    y, th0 = make_responses(75, 10, seed=14)
    ref = ShardedSampler(OracleEngine, y, th0, dist=None)
    ref.init()
    ref.step()
    assert np.array_equal(got["theta"], ref.engine.theta)
    assert np.abs(got["L"] - ref.engine.L).max() < 1e-12
    assert np.abs(np.triu(got["L"], 1)).max() == 0
    assert np.abs(got["f"] - ref.gather("f")).max() < 1e-9
    assert np.abs(got["fstar"] - ref.gather("fstar")).max() < 1e-9
