"""Two ranks sharing the one visible MI355X (gloo backend, device tensors): the item-sharded HIP sampler
must reproduce the single-process HIP sampler.  Exercises the real engine behind
gpirt_amd/distributed.py: zero-copy device views of sampler state, the all-reduce of the partial
log-posterior, the broadcast of L, the distributed factorisation (1-D block-cyclic outer panels, panel broadcasts
with one panel of look-ahead: L must be BIT-IDENTICAL to gpirt_sampler_factor), and the global-item RNG keys.  (RCCL itself needs >= 2 GPUs and is
exercised by bench.py --gpus N on the driver's 8-GPU node.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(chol):
    """(n, sampler options, ShardedSampler chol mode).  distributed: three 1024-column outer panels, the last ragged;
    distributed_lowrank: the bordered factorisation (64 extra rows riding through every panel and broadcast)."""
    if chol == "distributed":
        return 2500, {}, "distributed"
    if chol == "distributed_lowrank":
        return 2560, dict(fstar_fused=True, kstar_rank=64), "distributed"
    return 300, {}, chol


def _run(rank, world, port, chol, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpirt_amd.distributed import ShardedSampler
    from gpirt_amd.ops import Handle
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    n, kw, mode = _case(chol)
    y, th0 = make_responses(n, 22, seed=6)
    h = Handle(0)

    def factory(yl, th, pm, ps, st, item0, m_total):
        return Sampler(h, yl, th, pm, ps, st, rng="item", seed=77, item0=item0, m_total=m_total, **kw)

    ss = ShardedSampler(factory, y, th0, dist=dist, chol=mode)
    ss.init()
    for _ in range(2):
        ss.step()
    ss.engine.check()
    f, beta, fstar = ss.gather("f"), ss.gather("beta"), ss.gather("fstar")
    if rank == 0:
        np.savez(os.path.join(outdir, f"gpu_sharded_{chol}.npz"), f=f, beta=beta, fstar=fstar,
                 theta=ss.engine.get("theta"), L=ss.engine.get("L"))
    dist.destroy_process_group()


@pytest.mark.parametrize("chol", ["replicated", "bcast", "distributed", "distributed_lowrank"])
def test_two_ranks_one_gpu_match_single_process(handle, tmp_path, chol):
    import torch.multiprocessing as mp
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    port = 29700 + (os.getpid() % 1000) + {"replicated": 0, "bcast": 1, "distributed": 2, "distributed_lowrank": 3}[chol]
    mp.spawn(_run, args=(2, port, chol, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / f"gpu_sharded_{chol}.npz")
    n, kw, _ = _case(chol)
    y, th0 = make_responses(n, 22, seed=6)
    ref = Sampler(handle, y, th0, rng="item", seed=77, **kw)
    ref.init()
    for _ in range(2):
        ref.step()
    ref.check()
    assert np.array_equal(got["theta"], ref.get("theta"))
    assert np.abs(got["L"] - ref.get("L")).max() == 0
    assert np.abs(got["f"] - ref.get("f")).max() < 1e-10
    assert np.abs(got["beta"] - ref.get("beta")).max() < 1e-10
    assert np.abs(got["fstar"] - ref.get("fstar")).max() < 1e-10
    ref.close()


def _run_rccl(rank, world, port, chol, outdir):
    """One rank per GPU over RCCL (backend "nccl"): what bench.py --gpus N runs."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from gpirt_amd.distributed import ShardedSampler
    from gpirt_amd.ops import Handle
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(2560, 24, seed=6)
    h = Handle(rank)

    def factory(yl, th, pm, ps, st, item0, m_total):
        return Sampler(h, yl, th, pm, ps, st, rng="item", seed=77, item0=item0, m_total=m_total, fstar_fused=True,
                       kstar_rank=64)

    ss = ShardedSampler(factory, y, th0, dist=dist, chol=chol)
    ss.init()
    for _ in range(2):
        ss.step()
    ss.engine.check()
    f, beta, fstar = ss.gather("f"), ss.gather("beta"), ss.gather("fstar")
    if rank == 0:
        np.savez(os.path.join(outdir, f"rccl_{chol}.npz"), f=f, beta=beta, fstar=fstar, theta=ss.engine.get("theta"),
                 L=ss.engine.get("L"))
    dist.destroy_process_group()


@pytest.mark.parametrize("chol", ["replicated", "distributed"])
def test_two_gpus_over_rccl_match_one_gpu(handle, tmp_path, chol):
    """Needs two MI355X in the box (skipped on the one-GPU test box): item shards on two devices, the f* all-gather,
    the theta combine and -- for chol = distributed -- the panel broadcasts all travel over RCCL; theta and L must be
    bit-identical to the single-GPU run."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL parity needs two GPUs")
    import torch.multiprocessing as mp
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    port = 29900 + (os.getpid() % 1000) + (0 if chol == "replicated" else 1)
    mp.spawn(_run_rccl, args=(2, port, chol, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / f"rccl_{chol}.npz")
    y, th0 = make_responses(2560, 24, seed=6)
    ref = Sampler(handle, y, th0, rng="item", seed=77, fstar_fused=True, kstar_rank=64)
    ref.init()
    for _ in range(2):
        ref.step()
    ref.check()
    assert np.array_equal(got["theta"], ref.get("theta"))
    assert np.abs(got["L"] - ref.get("L")).max() == 0
    assert np.abs(got["f"] - ref.get("f")).max() < 1e-10
    assert np.abs(got["fstar"] - ref.get("fstar")).max() < 1e-10
    ref.close()
