#!/usr/bin/env python3
"""bench.py -- MCMC iterations/sec of the MI355X-native gpirtMCMC hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE full MCMC iteration of src/gpirtMCMC.cpp:68-78 on the device -- draw_f (RNG fill +
trmm + elliptical slice), draw_fstar (K*, trsm, [trsm^T], gemm, epilogue), draw_theta (MFMA GEMM +
grid inverse-CDF), draw_beta (+ mu, mu_star), K + jitter + blocked MFMA Cholesky -- on synthetic 2PL
responses of the metric's shape, N = 8192 respondents x m = 1024 items, fp64, item-keyed RNG
(GPIRT_RNG_ITEM).  Inputs are resident in HBM before the timed region.  With N > 1 the SAME problem is
sharded over item columns (strong scaling): for draw_theta the ranks all-gather their f* columns (8 MB) over
RCCL and each draws theta for its block of respondents (--theta allreduce: the 66 MB partial log-posteriors are
all-reduced instead); the Cholesky is replicated (--chol bcast: rank 0 factors and broadcasts L).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      the potrf trailing-update kernel (fp64 MFMA syrk): algorithmic flops / HIP-event time
                measured inside the timed region, against the 78.6 TFLOP/s fp64 matrix peak;
  cpu_baseline  the CPU oracle (a port of the reference; the reference itself needs R) timed on this
                host on a bounded sample of the same workload, one core.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6      # MI355X fp64 matrix peak (public spec; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)


def cpu_baseline(n, m, sampler, y, theta_host):
    """Time the oracle (port of the reference's per-item BLAS-2 structure) on a bounded sample."""
    import numpy as np
    from oracle import oracle as O          # checker / baseline only
    O.build()
    N = O.NGRID
    t = {}
    # K + chol on the leading n_s respondents, unblocked-order blocked code on ONE thread, scaled n^3
    n_s = min(n, 2048)
    t0 = time.perf_counter()
    _, info = O.factor(theta_host[:n_s], blocked=True, nthreads=1)
    t["chol"] = (time.perf_counter() - t0) * (n / n_s) ** 3
    L = sampler.get("L")
    f = sampler.get("f")
    beta = sampler.get("beta")
    mu = sampler.get("mu")
    ts = O.theta_star()
    mi = min(m, 2)
    rng = O.ItemStream(1)
    t0 = time.perf_counter()
    O.draw_f(rng, f[:, :mi], y[:, :mi], L, mu[:, :mi], it=1)
    t["draw_f"] = (time.perf_counter() - t0) * (m / mi)
    # draw_fstar: the item-independent trsm over a sample of grid columns + per-item double solves
    gs = 8
    mu_star = beta[0][None, :mi] + ts[:gs, None] * beta[1][None, :mi]
    t0 = time.perf_counter()
    O.draw_fstar(rng, f[:, :mi], theta_host, L, mu_star, it=1, tstar=ts[:gs])
    dt = time.perf_counter() - t0
    # split: common part scales with N/gs, per-item part with m/mi (both are O(n^2) solves)
    common_share = gs / (gs + 2.0 * mi)
    t["draw_fstar"] = dt * common_share * (N / gs) + dt * (1 - common_share) * (m / mi)
    fstar = sampler.get("fstar")
    ns = min(n, 4)
    t0 = time.perf_counter()
    O.draw_theta(rng, y[:ns, :], fstar, it=1, stabilise=True)
    t["draw_theta"] = (time.perf_counter() - t0) * (n / ns)
    pm, ps, st = np.zeros((2, mi)), np.full((2, mi), 3.0), np.full((2, mi), 0.1)
    t0 = time.perf_counter()
    O.draw_beta(rng, beta[:, :mi], theta_host, y[:, :mi], f[:, :mi], pm, ps, st, it=1)
    t["draw_beta"] = (time.perf_counter() - t0) * (m / mi)
    total = sum(t.values())
    return {
        "value": 1.0 / total, "unit": "iterations/s", "cores": 1, "kind": "port",
        "sample": (f"oracle (C restatement of the reference) on 1 thread, extrapolated from: chol on the leading "
                   f"{n_s} respondents (x(n/{n_s})^3), draw_f/draw_beta on {mi} of {m} items, draw_fstar on {gs} of "
                   f"{N} grid columns + {mi} items, draw_theta on {ns} of {n} respondents"),
        "stage_seconds": {k: round(v, 3) for k, v in t.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--respondents", "--n", dest="n", type=int, default=8192)
    ap.add_argument("--items", "--m", dest="m", type=int, default=1024)
    ap.add_argument("--chol", default="replicated", choices=["replicated", "bcast"])
    ap.add_argument("--theta", default="gather", choices=["gather", "allreduce"],
                    help="N > 1: all-gather f* and draw theta per respondent block (default), or all-reduce the partial log-posterior")
    ap.add_argument("--fstar", default="lowrank", choices=["double_solve", "fused", "lowrank"],
                    help="double_solve: src/draw-fstar.cpp as written; fused: mean = (L^-1 k*)^T (L^-1 f); lowrank: fused + the "
                         "rank-64 Chebyshev factorisation of K(theta, theta*) (exact to 1e-15), 64 + m right-hand sides")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-forms", action="store_true",
                    help="skip the extra (untimed for `value`) runs with the other draw_fstar forms")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo; RCCL refuses duplicate devices)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from gpirt_amd.distributed import ShardedSampler
    from gpirt_amd.ops import Handle
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses

    n, m = args.n, args.m
    y, theta0 = make_responses(n, m, seed=20240)
    handle = Handle(local_rank)

    def factory(y_loc, th, pm, ps, st, item0, m_total):
        return Sampler(handle, y_loc, th, pm, ps, st, rng="item", seed=20240, theta_stabilise=True,
                       fstar_fused=(args.fstar != "double_solve"), kstar_rank=(64 if args.fstar == "lowrank" else 0),
                       item0=item0, m_total=m_total)

    ss = ShardedSampler(factory, y, theta0, dist=dist if world > 1 else None, chol=args.chol, theta=args.theta)
    ss.init()
    ss.engine.check()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ss.step()
    ss.engine.check()
    handle.prof_trailing(reset=True)
    handle.prof_enable(True)          # event pairs around each trailing-update launch, no host sync
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ss.step()
    barrier()
    dt = time.perf_counter() - t0
    handle.prof_enable(False)
    ss.engine.check()
    tr_ms, tr_launches, tr_flops = handle.prof_trailing(reset=True)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-stage device times of two extra (untimed) iterations, device events on the launch stream
    evs = []

    def timer(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        evs.append((name, ev))

    stage_ms = {}
    for rep in range(2):
        evs.clear()
        barrier()
        timer("start")
        ss.step(timer)
        torch.cuda.synchronize()
        cur = {evs[i][0]: evs[i - 1][1].elapsed_time(evs[i][1]) for i in range(1, len(evs))}
        stage_ms = cur if not stage_ms else {k: min(stage_ms[k], cur[k]) for k in cur}
    ss.engine.check()
    stage_ms = {k: round(v, 3) for k, v in stage_ms.items()}

    # the same iteration with draw_fstar as the reference words it (every one of the 1001 grid columns solved,
    # then L^-T L^-1 f per item): reported beside `value`, never as it
    alt = None
    if world == 1 and args.fstar == "lowrank" and not args.no_alt_forms:
        alt = {}
        for form, kw in (("fused", dict(fstar_fused=True)), ("double_solve", dict(fstar_fused=False))):
            s2 = Sampler(handle, y, theta0, rng="item", seed=20240, theta_stabilise=True, **kw)
            s2.init()
            for _ in range(max(1, args.warmup)):
                s2.step()
            s2.check()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                s2.step()
            torch.cuda.synchronize()
            alt[form] = round(args.steps / (time.perf_counter() - t1), 3)
            s2.check()
            s2.close()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        achieved = (tr_flops / (tr_ms * 1e-3) / 1e12) if tr_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "trailing_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "MCMC iterations/sec at N=8192 x m=1024" if (n, m) == (8192, 1024)
                      else f"MCMC iterations/sec at N={n} x m={m} (not the BASELINE metric shape)",
            "value": args.steps / dt,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"M: N={n} respondents x m={m} items, synthetic 2PL responses (5% NA), full MCMC "
                            f"iteration on device (draw_f, draw_fstar[{args.fstar}], draw_theta, draw_beta, K+chol), "
                            f"rng=item, theta_stabilise=1",
                "parallelism": (f"items sharded over {world} GPU(s); chol {args.chol}; draw_theta: " +
                                (f"all-gather of f* ({1001}x{m}), theta drawn per block of respondents, {n} draws combined"
                                 if args.theta == "gather" else
                                 f"all-reduce of the {1001}x{n} partial log-posterior per iteration")) if world > 1 else "single GPU",
                "stage_ms": stage_ms,
                "draw_fstar_form": {"double_solve": "src/draw-fstar.cpp:17-25 as written",
                                    "fused": "mean = (L^-1 k*)^T (L^-1 f)",
                                    "lowrank": "fused + K(theta, theta*) = K(theta, c) V^T, 64 Chebyshev nodes (max-abs error "
                                               "1.3e-15; f* within 2e-11 of the full solve): 2 x 64 right-hand sides instead "
                                               "of 1001 + m"}[args.fstar],
                "iterations_per_s_other_forms": alt,
                "item_sharded_stages": ["draw_f", "draw_fstar", "draw_beta"] + (["theta_gemm"] if args.theta == "allreduce" else []),
                "respondent_sharded_stages": ["theta_gemm", "theta_sample"] if (world > 1 and args.theta == "gather") else [],
                "stage_ms_note": "theta_allreduce = the collective of draw_theta (all-gather of f* or all-reduce of the log-posterior)",
            },
            "roofline": {
                "kernel": ("gemm_f64_kernel<false, true, 64, 0, false> (potrf trailing update, deferred block columns, syrk lower, v_mfma_f64_16x16x4_f64)"
                           if os.environ.get("GPIRT_DEFER") in ("1", "3") else
                           "gemm_f64_kernel<false, true, 128, 8, false> (potrf trailing update, syrk lower, v_mfma_f64_16x16x4_f64)"),
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                "traffic": traffic,
                "launches": int(tr_launches),
                "avg_launch_ms": (tr_ms / tr_launches) if tr_launches else None,
                "flops_per_launch": (tr_flops / tr_launches) if tr_launches else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, m, ss.engine, y, ss.engine.get("theta"))
            except Exception as e:      # the baseline is a reported extra; never lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
