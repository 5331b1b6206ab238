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
all-reduced instead); the Cholesky is replicated (--chol bcast: rank 0 factors and broadcasts L; --chol distributed:
block-cyclic outer panels, each finished panel broadcast while the previous one is applied).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      every syrk launch of the factorisation (fp64 MFMA, both tile instantiations): algorithmic flops /
                HIP-event time measured inside the timed region, against the 78.6 TFLOP/s fp64 matrix peak;
  cpu_baseline  the CPU oracle (a port of the reference; the reference itself needs R) timed on this
                host on a bounded sample of the same workload: one thread (reference-shaped) and all cores;
  config.lowrank_check  max|f*_lowrank - f*_full-solve| measured in this run on the state the timed steps ended in
                (the rank-64 form of draw_fstar is opt-in; if it misses 1e-9 the headline is re-timed with `fused`);
  config.reference_rng_iterations_per_s  (N = 1) the rate of the literal gpirt_default_options contract on the same
                problem -- R-stream replay, src/draw-fstar.cpp as written, draw_theta as written: what an R user of the
                unmodified shim gets -- measured over 2 iterations outside the timed region;
  item_shard_speedup, whole_iteration_speedup, rccl_ranks  (N > 1) the item-sharded stages and the whole iteration
                against a single-GPU run of the FULL problem that rank 0 times in the same process, same K and W.
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
PEAK_INT8_MFMA_TOPS = 5000.0      # dense int8 matrix peak: twice the bf16 rate per clock (MI355X_MICROARCH.md: ~2.5 PF bf16 dense)
PEAK_HBM_GBS = 8000.0             # HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, 6.29 TB/s measured for a copy)


def cpu_baseline(n, m, sampler, y):
    """the CPU restatement of the reference on this host.  Top-level value: ALL host cores, one whole
    iteration at the full size (blocked OpenMP potrf + item-parallel stages), measured, nothing extrapolated; the
    reference-shaped single-thread figure (unblocked potrf, stretched from samples) is the sub-object
    `single_thread_reference_shaped`."""
    from oracle import cpu_baseline as CB          # checker / baseline only; never the product path
    return CB.run(n, m, y, sampler.get("theta"), sampler.get("L"), sampler.get("f"), sampler.get("beta"),
                  sampler.get("mu"), sampler.get("fstar"))


FSTAR_TOL = 1e-9       # tolerance of every f* comparison (north star: posterior means within 1e-8 relative)


def _roof_entry(prof, bound, peak, unit, kernel):
    """One more roofline entry from the library's event pairs: (ms, launches, flops, bytes) of a kernel class."""
    if not prof or not prof[1] or prof[0] <= 0:
        return None
    ms, nl, fl, by = prof
    work = fl if bound == "mfma" else by
    achieved = work / (ms * 1e-3) / (1e12 if bound == "mfma" else 1e9)
    return {"kernel": kernel, "bound": bound, "launches": int(nl), "avg_launch_ms": ms / nl,
            "flops_per_launch": fl / nl, "algorithmic_bytes_per_launch": by / nl,
            "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak}


def _replay_entry(prof, n):
    """The roofline entry of the R-stream predictor's products kernel.  From 4096 respondents on the pass is STRUCTURED
    (csrc/rs_lr.hip): it reads the 512-column diagonal parts of L and 64 rows of coefficients instead of the lower triangle --
    told apart here by the bytes the library booked per launch."""
    e = _roof_entry(prof, "hbm", PEAK_HBM_GBS, "GB/s", "")
    if e is None:
        return None
    dense_bytes = 4.0 * 0.5 * n * (n + 1)
    if e["algorithmic_bytes_per_launch"] < 0.5 * dense_bytes:
        e["kernel"] = ("rs3p_products_kernel (R-stream replay, gpirt_default_options), STRUCTURED pass (csrc/rs_lr.hip): the blocks of L below "
                       "the 512-column diagonal parts are applied as V C (Lagrange basis of 64 Chebyshev nodes at theta x coefficients "
                       "built from theta alone), so a pass reads 4 (512 + 64) n bytes -- %.1f MB instead of the lower triangle's %.1f MB as "
                       "floats -- and this kernel is bound by its launch and two memory round trips, not by bytes; beside it per pass: "
                       "rs_lr_apply_kernel (prefix over the parts' records + V x prefix) and rs3p_decide_kernel.  GPIRT_RS_LR=2: the dense "
                       "single-precision pass (134 MB at n = 8192: 30 us = 0.52-0.57 of the HBM peak, profiles/r06_replay_dense_summary.md)"
                       % (e["algorithmic_bytes_per_launch"] / 1e6, dense_bytes / 1e6))
        e["structured"] = True
    else:
        e["kernel"] = ("rs3p_products_kernel (R-stream replay, gpirt_default_options): the predictor's pass over L as single-precision "
                       "tiles; bytes = the lower triangle as floats, 4 n (n + 1) / 2, per launch (the spare passes that find every item "
                       "predicted and leave at once are left out, as in rocprofv3's median in profiles/); flops = "
                       "2 x 32 candidates x n (n + 1) / 2 on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak: 16.6 us of a ~30 us pass)")
        e["structured"] = False
    e["kernel"] += "; timed in the reference_rng leg of this run"
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--respondents", "--n", dest="n", type=int, default=8192)
    ap.add_argument("--items", "--m", dest="m", type=int, default=1024)
    ap.add_argument("--chol", default="replicated", choices=["replicated", "bcast", "distributed"],
                    help="N > 1: every rank factors (default); rank 0 factors and broadcasts L; or the factorisation is "
                         "distributed: 1-D block-cyclic outer panels, panel broadcasts overlapped with the trailing updates")
    ap.add_argument("--theta", default="gather", choices=["gather", "allreduce"],
                    help="N > 1: all-gather f* and draw theta per respondent block (default), or all-reduce the partial log-posterior")
    ap.add_argument("--fstar", default="lowrank", choices=["double_solve", "fused", "lowrank"],
                    help="double_solve: src/draw-fstar.cpp as written; fused: mean = (L^-1 k*)^T (L^-1 f); lowrank: fused + the "
                         "rank-64 Chebyshev factorisation of K(theta, theta*) (exact to 1e-15), 64 + m right-hand sides")
    ap.add_argument("--kernel-fp32", action="store_true",
                    help="BASELINE config C5: build K(theta, theta) in single precision (gpirt_options.kernel_fp32), factor in fp64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-rng", action="store_true",
                    help="skip the (untimed for `value`) two iterations under the literal default contract (R-stream replay)")
    ap.add_argument("--no-single-gpu-reference", action="store_true",
                    help="N > 1: skip rank 0's single-GPU run of the full problem (item_shard_speedup / whole_iteration_speedup)")
    ap.add_argument("--no-alt-forms", action="store_true",
                    help="skip the extra (untimed for `value`) runs with the other draw_fstar forms")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo; RCCL refuses duplicate devices). The persistent "
                         "sub-panel kernel needs its work-groups co-resident (DESIGN.md section 4): two processes on one card "
                         "still get that, four do not (the hang guard fires and the run fails loudly), so with more than two "
                         "ranks the rehearsal factors with the launch-per-step panel (GPIRT_PANEL=2)")
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
        if world > 2:
            os.environ.setdefault("GPIRT_PANEL", "2")     # (read once, when the library is loaded below)
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
    FORMS = {"double_solve": dict(fstar_fused=False, kstar_rank=0), "fused": dict(fstar_fused=True, kstar_rank=0),
             "lowrank": dict(fstar_fused=True, kstar_rank=64)}

    def make(form, single=False):
        def factory(y_loc, th, pm, ps, st, item0, m_total):
            if form == "lowrank":       # the headline: the library's throughput preset, gpirt_fast_options() (include/gpirt_hip.h)
                return Sampler(handle, y_loc, th, pm, ps, st, preset="fast", seed=20240, item0=item0, m_total=m_total,
                               kernel_fp32=args.kernel_fp32)
            return Sampler(handle, y_loc, th, pm, ps, st, rng="item", seed=20240, theta_stabilise=True,
                           item0=item0, m_total=m_total, kernel_fp32=args.kernel_fp32, **FORMS[form])
        if single:      # the FULL problem on this rank alone (rank 0's single-GPU reference inside a sharded run)
            return ShardedSampler(factory, y, theta0, dist=None)
        return ShardedSampler(factory, y, theta0, dist=dist if world > 1 else None, chol=args.chol, theta=args.theta)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    prof_other = {}

    def timed_run(ss, local=False):
        """W warm-up steps, then exactly K timed steps between barrier + synchronize; max over ranks.
        local: this rank alone (no barrier, no reduction): rank 0's single-GPU reference run."""
        if local:
            for _ in range(args.warmup):
                ss.step()
            ss.engine.check()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ss.step()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0
            ss.engine.check()
            return dt_, None
        for _ in range(args.warmup):
            ss.step()
        ss.engine.check()
        handle.prof_syrk(reset=True)
        handle.prof_other(reset=True)
        # The roofline events are an instrument with a price: a pair of hipEventRecord around each of the 54 syrk
        # launches of a step costs 4 % of the iteration rate when every timed step carries them (124.3 against 129.6 it/s).
        # So they ride on a sample of the timed steps -- the first and the middle one -- which is still a measurement
        # inside the timed region, over the launches of whole factorisations.
        sample = {0, args.steps // 2} if os.environ.get("BENCH_PROF_ALL") != "1" else set(range(args.steps))
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            handle.prof_enable(i in sample)   # host-side flag: event pairs around each syrk launch, no host sync
            ss.step()
        barrier()
        dt = time.perf_counter() - t0
        handle.prof_enable(False)
        ss.engine.check()
        prof = handle.prof_syrk(reset=True)
        prof_other.clear(); prof_other.update(handle.prof_other(reset=True))       # (nu = L Z of draw_f on the sampled steps)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, prof

    form = args.fstar
    ss = make(form)
    ss.init()
    ss.engine.check()
    dt, prof = timed_run(ss)

    # ---- self-certification of the opt-in low-rank form, in this run, on the state the timed steps ended in: the same
    # draw_fstar (same theta, f, L, mu_star, same RNG keys) with every one of the 1001 grid columns solved
    # (src/draw-fstar.cpp:17-25 in its `fused` wording) on a second sampler; outside the timed region.
    lowrank_gap = None
    headline_note = None
    if form == "lowrank":
        e = ss.engine
        chk = make("fused")
        chk.init()
        c = chk.engine
        c.copy_state_from(e)           # theta, f, beta, mu, mu_star, L, iteration counter (device to device)
        e.draw_fstar()
        c.draw_fstar()
        e.check(); c.check()
        gap = (e.device_tensor("fstar") - c.device_tensor("fstar")).abs().max()
        # ... and against src/draw-fstar.cpp:17-25 AS WRITTEN (two solves per item, mean = k*^T alpha) on a third sampler
        aw = make("double_solve")
        aw.init()
        w = aw.engine
        w.copy_state_from(e)
        w.draw_fstar()
        w.check()
        gap_aw = (e.device_tensor("fstar") - w.device_tensor("fstar")).abs().max()
        fs_scale = w.device_tensor("fstar").abs().max()
        if world > 1:
            dist.all_reduce(gap, op=dist.ReduceOp.MAX)
            dist.all_reduce(gap_aw, op=dist.ReduceOp.MAX)
            dist.all_reduce(fs_scale, op=dist.ReduceOp.MAX)
        lowrank_gap = float(gap.item())
        lowrank_gap_aw = float(gap_aw.item())
        fstar_scale = max(1.0, float(fs_scale.item()))
        del aw, w
        theta_on_grid = bool(torch.all(((e.device_tensor("theta") + 5.0) / 0.01 - torch.round((e.device_tensor("theta") + 5.0) / 0.01)).abs() < 1e-9).item())
        # (both comparisons relative to max|f*|, as in tests/test_gpu_configs.py: two backward-stable evaluations of k*^T S^-1 f
        # differ by ~cond(S) eps |f*| -- at n = 16384 the absolute 1e-9 sits AT that floor, 0.8e-9 ... 1.02e-9 from state to state)
        if not (lowrank_gap <= FSTAR_TOL * fstar_scale and lowrank_gap_aw <= FSTAR_TOL * fstar_scale):
            # the low-rank form missed the tolerance on this state: the headline falls back to the like-for-like form
            headline_note = (f"lowrank measured max|f*_lowrank - f*_fused| = {lowrank_gap:.3e} (tolerance {FSTAR_TOL:g} x max|f*| = {FSTAR_TOL * fstar_scale:.3e}) and "
                             f"max|f*_lowrank - f*_as_written| = {lowrank_gap_aw:.3e} (tolerance {FSTAR_TOL:g} x max|f*| = "
                             f"{FSTAR_TOL * fstar_scale:.3e}): `value` is the `fused` form")
            form = "fused"
            ss = chk
            dt, prof = timed_run(ss)
        else:
            del chk, c

    # ---- the log-posterior product of draw_theta on the state the timed steps ended in, both ways: in fixed point on the
    # int8 matrix cores (what the timed steps ran, csrc/theta_fixed.hip) and as the fp64 GEMM it replaces (GPIRT_THETA_FIXED=2)
    theta_product_check = None
    if world == 1:
        from gpirt_amd.ops import to_device
        e = ss.engine
        yd = to_device(y)
        fd = e.device_tensor("fstar")
        with handle.config("GPIRT_THETA_FIXED", 1):
            lp_fx, fell_back = handle.theta_logpost(yd, fd)
        with handle.config("GPIRT_THETA_FIXED", 2):
            lp_ge, _ = handle.theta_logpost(yd, fd)
        theta_product_check = {
            "max_abs_fixed_minus_fp64_gemm": float((lp_fx - lp_ge).abs().max().item()),
            "max_abs_logpost": float(lp_ge.abs().max().item()),
            "fixed_point_handed_over_to_fp64": bool(fell_back),
            "what": "N x n log-posterior of draw_theta from this run's last f*: seven int8 digit planes with exact int32 sums (every "
                    "term rounded once to 54 bits of its grid row's range) against the fp64 MFMA GEMM; tests/test_gpu_theta_fixed.py "
                    "holds the fixed-point form to the SMALLER error against long double",
        }
        del lp_fx, lp_ge, yd

    # per-stage device times of two extra (untimed) iterations, device events on the launch stream
    evs = []

    def timer(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        evs.append((name, ev))

    def stage_times(sx, sync=barrier):
        best = {}
        for rep in range(2):
            evs.clear()
            sync()
            timer("start")
            sx.step(timer)
            torch.cuda.synchronize()
            cur = {evs[i][0]: evs[i - 1][1].elapsed_time(evs[i][1]) for i in range(1, len(evs))}
            best = cur if not best else {k: min(best[k], cur[k]) for k in cur}
        sx.engine.check()
        return {k: round(v, 3) for k, v in best.items()}

    stage_ms = stage_times(ss)
    # every rank's stage times in rank 0's line: a first multi-GPU run is then diagnosable from one log (a slow rank, a
    # collective that waits, the Cholesky mode in use)
    stage_ms_per_rank = None
    if world > 1:
        stage_ms_per_rank = [None] * world
        dist.all_gather_object(stage_ms_per_rank, stage_ms)        # a few hundred bytes, outside every timed region

    # ---- N > 1: the north star's "item-shard speed-up 1 -> N" as ONE command.  Rank 0 runs the FULL problem alone on its
    # GPU (the other ranks wait at the barrier, their GPUs idle), same form, same K and W, outside the timed region: stage
    # times for the item-sharded stages and a timed run for the whole iteration.
    single = None
    if world > 1 and not args.no_single_gpu_reference:
        if rank == 0:
            s1 = make(form, single=True)
            s1.init()
            s1.engine.check()
            dt1, _ = timed_run(s1, local=True)
            st1 = stage_times(s1, sync=torch.cuda.synchronize)
            single = {"iterations_per_s": args.steps / dt1, "stage_ms": st1}
            del s1
        barrier()

    # ---- N = 1: what the DEFAULT contract costs (gpirt_default_options: R-stream replay, src/draw-fstar.cpp as written,
    # draw_theta as written).  The replay is item-sequential by construction -- item j's normals start where item j - 1's
    # data-dependent slice loop stopped consuming (src/draw-f.cpp:40-58) -- so this is m dependent products per iteration.
    ref_rng = None
    replay_prof = None
    if world == 1 and not args.no_reference_rng:
        from gpirt_amd.ops import RStream
        literal_error = None
        for stab in (False, True):
            # The LITERAL default first.  At many items draw_theta as written underflows (exp of a sum of ~m log-likelihood
            # terms: 0/0 for some respondents, where the reference reads theta_star[N] out of bounds -- quirk Q5); the run
            # then says so and the rate is measured with theta_stabilise = 1, the same draw wherever the reference is defined.
            sr = Sampler(handle, y, theta0, rng="reference", rstream=RStream(20240), theta_stabilise=stab, fstar_fused=False)
            try:
                sr.init()
                sr.check()
                sr.step()                       # (first iteration: workspaces)
                sr.check()
                torch.cuda.synchronize()
                handle.prof_other(reset=True)
                t0 = time.perf_counter()
                for it_ in range(2):
                    handle.prof_enable(it_ == 1)        # event pairs around the second iteration's passes over L
                    sr.step()
                torch.cuda.synchronize()
                dtr = time.perf_counter() - t0
                handle.prof_enable(False)
                sr.check()
                replay_prof = handle.prof_other(reset=True)["replay_products"]
                rs_stats = sr.get("rs_stats")
                ref_rng = {"value": 2.0 / dtr, "iterations": 2, "theta_stabilise": int(stab),
                           "passes_over_L_per_iteration": int(replay_prof[1]), "items_per_pass": (m / replay_prof[1]) if replay_prof[1] else None,
                           "draw_f": "predict + verify (csrc/rs_predict.hip, csrc/rs_lr.hip): the starts of all items in R's stream predicted by passes over a "
                                     "SINGLE-PRECISION copy of L (up to four items per pass, 32 candidate starts, 16 trial points each; from 4096 respondents on a "
                                     "pass reads only L's 512-column diagonal parts and applies the blocks below them in a rank-64 Lagrange basis), then every item "
                                     "computed exactly at its predicted start -- one fp64 triangular MFMA product + all slice loops side by side, the "
                                     "formula as written -- and committed in order; GPIRT_RS_PREDICT=2: every pass in fp64 (rounds 4-5)",
                           "mispredictions_found_by_the_verification_so_far": int(rs_stats[1]),
                           "predictor_stalls_handed_to_the_one_phase_replay": int(rs_stats[2]),
                           "contract": "gpirt_default_options: rng = R-stream replay (item-sequential draw_f), draw_fstar = double_solve "
                                       "as written" + ("; theta_stabilise = 1 because the literal default (0) failed on this problem: "
                                                       + literal_error if stab else "; theta_stabilise = 0 (the literal default)")}
                sr.close()
                # the same contract's RNG -- R's stream, consumed exactly as the reference consumes it -- with the cheaper,
                # algebraically identical forms of draw_fstar (R: options(gpirt.hip.fstar_fused = TRUE, gpirt.hip.kstar_rank = 64))
                by_form = {"double_solve": ref_rng["value"]}
                for name_, kw_ in (("fused", dict(fstar_fused=True)), ("lowrank", dict(fstar_fused=True, kstar_rank=64))):
                    s2 = Sampler(handle, y, theta0, rng="reference", rstream=RStream(20240), theta_stabilise=stab, **kw_)
                    try:
                        s2.init(); s2.check(); s2.step(); s2.check()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(2):
                            s2.step()
                        torch.cuda.synchronize()
                        by_form[name_] = 2.0 / (time.perf_counter() - t0)
                        s2.check()
                    except Exception as exc2:
                        by_form[name_] = None
                        by_form[name_ + "_error"] = repr(exc2)
                    s2.close()
                ref_rng["iterations_per_s_by_fstar_form"] = by_form
                break
            except Exception as exc:
                literal_error = repr(exc)
                ref_rng = {"value": None, "iterations": 2, "theta_stabilise": int(stab), "contract": "gpirt_default_options",
                           "error": literal_error}
                sr.close()

    # the same iteration with draw_fstar in the other forms (one GPU only), same K and W: reported beside `value`
    alt = None
    if world == 1 and not args.no_alt_forms:
        alt = {form: round(args.steps / dt, 3)}
        for f2 in ("lowrank", "fused", "double_solve"):
            if f2 in alt:
                continue
            s2 = make(f2)
            s2.init()
            dt2, _ = timed_run(s2)
            alt[f2] = round(args.steps / dt2, 3)
            del s2

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        tot_ms = sum(v[0] for v in prof.values())
        tot_fl = sum(v[2] for v in prof.values())
        tot_n = sum(v[1] for v in prof.values())
        achieved = (tot_fl / (tot_ms * 1e-3) / 1e12) if tot_ms > 0 else 0.0
        tot_by = sum(v[3] for v in prof.values())
        n_sampled = args.steps if os.environ.get("BENCH_PROF_ALL") == "1" else len({0, args.steps // 2})
        by_class = {k: {"launches": int(v[1]), "avg_launch_ms": (v[0] / v[1]) if v[1] else None,
                        "flops_per_launch": (v[2] / v[1]) if v[1] else None,
                        "algorithmic_bytes_per_launch": (v[3] / v[1]) if v[1] else None,
                        "achieved": (v[2] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else None,
                        "frac": (v[2] / (v[0] * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS) if v[0] > 0 else None,
                        "ms_per_step": v[0] / n_sampled}
                    for k, v in prof.items()}
        traffic, traffic_src, mfma_busy = None, None, None
        tpath = os.path.join(ROOT, "profiles", "trailing_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), ("STATIC FILE profiles/trailing_traffic.json (PMC passes of an "
                                                                         "earlier run of this command, not this run): " + str(tj.get("source")))
                mfma_busy = {k: v.get("mfma_busy_frac") for k, v in (tj.get("by_kernel") or {}).items()}
            except Exception:
                traffic = None
        sharded = ["draw_f", "draw_beta"] + (["theta_gemm"] if args.theta == "allreduce" else [])
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
                            f"iteration on device (draw_f, draw_fstar[{form}], draw_theta, draw_beta, K+chol), "
                            f"rng=item, theta_stabilise=1" +
                            ("; options = gpirt_fast_options() (include/gpirt_hip.h: the library's throughput preset -- item-keyed "
                             "RNG, theta_stabilise, fused + rank-64 draw_fstar); draw_fstar[lowrank] is checked in this run against the "
                             "full solve and the as-written form (lowrank_check)" if form == "lowrank" else ""),
                "options_preset": "gpirt_fast_options" if form == "lowrank" else None,
                "kernel_fp32": bool(args.kernel_fp32),
                "parallelism": (f"items sharded over {world} GPU(s); chol {args.chol}; draw_theta: " +
                                (f"all-gather of f* ({1001}x{m}), theta drawn per block of respondents, {n} draws combined"
                                 if args.theta == "gather" else
                                 f"all-reduce of the {1001}x{n} partial log-posterior per iteration")) if world > 1 else "single GPU",
                "stage_ms": stage_ms,
                "stage_ms_per_rank": stage_ms_per_rank,
                "chol": args.chol if world > 1 else "single GPU",
                "theta": args.theta if world > 1 else "single GPU",
                "backend": args.backend if world > 1 else None,
                "draw_fstar_form": {"double_solve": "src/draw-fstar.cpp:17-25 as written",
                                    "fused": "mean = (L^-1 k*)^T (L^-1 f)",
                                    "lowrank": "fused + K(theta, theta*) = K(theta, c) V^T, 64 Chebyshev nodes: 2 x 64 right-hand "
                                               "sides instead of 1001 + m"}[form],
                "lowrank_check": None if lowrank_gap is None else {
                    "max_abs_fstar_lowrank_minus_full_solve": lowrank_gap, "tolerance": FSTAR_TOL * fstar_scale,
                    "max_abs_fstar_lowrank_minus_as_written": lowrank_gap_aw,
                    "tolerance_as_written": FSTAR_TOL * fstar_scale, "max_abs_fstar": fstar_scale,
                    "as_written": "src/draw-fstar.cpp:17-25 (double_solve) on the same state and RNG keys; tolerance 1e-9 x max|f*| "
                                  "(two backward-stable evaluations of k*^T S^-1 f differ by ~cond(S) eps |f*|, DESIGN.md section 5)",
                    "state": f"after {args.warmup + args.steps} iterations of this run, theta on the grid: {theta_on_grid}",
                    "passed": bool(lowrank_gap <= FSTAR_TOL * fstar_scale and lowrank_gap_aw <= FSTAR_TOL * fstar_scale)},
                "theta_product": "exact fixed point on the int8 matrix cores (csrc/theta_fixed.hip; GPIRT_THETA_FIXED=2: fp64 GEMM)",
                "theta_product_check": theta_product_check,
                "headline_note": headline_note,
                "iterations_per_s_by_form": alt,
                "reference_rng_iterations_per_s": None if ref_rng is None else ref_rng["value"],
                "reference_rng": ref_rng,
                "item_sharded_stages": sharded + ([] if form == "lowrank" else ["draw_fstar (item part)"]),
                "replicated_stages": ["factor"] + (["draw_fstar: block inverses of L and the two 64-column solves (per-item part: "
                                                    "64 x m products + epilogue, sharded)"] if form == "lowrank"
                                                   else ["draw_fstar: L^-1 k* over the 1001 grid columns"]),
                "respondent_sharded_stages": ["theta_gemm", "theta_sample"] if (world > 1 and args.theta == "gather") else [],
                "stage_ms_note": "theta_allreduce = the collective of draw_theta (all-gather of f* or all-reduce of the log-posterior)",
            },
            "roofline": {
                "kernel": "gemm_f64_kernel<false, true, T, 8, false>, T = 128 and 64: every syrk-lower launch of the factorisation "
                          "(trailing updates + the update between the two sub-panels of each outer panel, K = the first sub-panel's width: 704 at this size, class `in_panel_update`), v_mfma_f64_16x16x4_f64",
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                "traffic": traffic,
                "traffic_source_static_pmc_file": traffic_src,
                "mfma_busy_static_pmc_file": mfma_busy,
                "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES share of the chip's SIMDs) of these launches, from the same "
                                  "static PMC file: the share of cycles the matrix pipes are busy.  `frac` above is against the NOMINAL peak "
                                  "(2.4 GHz); under this load the chip holds ~2.1 GHz (shader-cycle counter, DESIGN.md section 4), where the "
                                  "MFMA ceiling is 68.8 TFLOP/s",
                "launches": int(tot_n),
                "avg_launch_ms": (tot_ms / tot_n) if tot_n else None,
                "flops_per_launch": (tot_fl / tot_n) if tot_n else None,
                "algorithmic_bytes_per_launch": (tot_by / tot_n) if tot_n else None,
                "traffic_over_algorithmic": (traffic / (tot_by / tot_n)) if (traffic and tot_n and tot_by) else None,
                "by_class": by_class,
                # the whole factorisation (K + chol): n^3 / 3 flops against the stage's device time -- pivot chain, panels and all
                "factor_overall": {"bound": "mfma", "flops": n ** 3 / 3.0, "stage_ms": stage_ms.get("factor"),
                                   "achieved": (n ** 3 / 3.0 / (stage_ms["factor"] * 1e-3) / 1e12) if stage_ms.get("factor") else None,
                                   "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": (n ** 3 / 3.0 / (stage_ms["factor"] * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS) if stage_ms.get("factor") else None},
                # nu = L Z of draw_f (src/mvnormal.h:10 for all m columns as ONE triangular product, gemm_f64_kernel<false, false, ...>)
                "draw_f_trmm": _roof_entry(prof_other.get("draw_f_trmm"), "mfma", PEAK_FP64_MFMA_TFLOPS, "TFLOP/s",
                                           "gemm_f64_kernel<false, false, 128, 0, false>: n^2 m flops (the zero triangle skipped)"),
                # draw_theta's log-posterior product in fixed point: seven int8 digit planes against the 0/1 indicators
                "theta_int8_product": _roof_entry(prof_other.get("theta_int8_product"), "mfma", PEAK_INT8_MFMA_TOPS, "TOP/s",
                                                  "tf_mfma_kernel (csrc/theta_fixed.hip), v_mfma_i32_32x32x32_i8: 7 x 2 x 1001 x n x 2m integer "
                                                  "operations per launch, exact int32 sums; peak = twice the dense bf16 rate "
                                                  "(MI355X_MICROARCH.md, matrix cores) at the nominal 2.4 GHz -- the chip holds ~1.8 GHz under this "
                                                  "kernel (tools/theta_clock.py).  The fp64 GEMM it replaces ran 0.52 ms at 0.91 of the fp64 MFMA peak"),
                # the default contract's draw_f: the predictor's pass for up to four items (rs3p_products_kernel)
                "replay_products": _replay_entry(replay_prof, n),
                "note": "HIP events around each launch on its own stream (main or look-ahead side stream) inside the timed region, on "
                        "a sample of the timed steps (the first and the middle one: bracketing every launch of every step costs 4 % "
                        "of the iteration rate); launches of the two streams overlap each other and the panel kernel, so "
                        "per-launch times include that contention",
                "steps_sampled": n_sampled,
            },
        }
        if world > 1:
            # speed-ups against rank 0's single-GPU run of the full problem in this same process (None when it was skipped)
            out["rccl_ranks"] = dist.get_world_size()
            out["backend_reported"] = dist.get_backend()
            if single is not None:
                sh = [k for k in ("draw_f", "draw_fstar", "draw_beta") if k in single["stage_ms"]]
                worst = {k: max(r[k] for r in stage_ms_per_rank) for k in sh}          # the slowest rank sets the pace
                per = {k: (single["stage_ms"][k] / worst[k]) if worst[k] > 1e-3 else None for k in sh}
                tot1 = sum(single["stage_ms"][k] for k in sh)
                totN = sum(worst[k] for k in sh)
                out["item_shard_speedup"] = {"per_stage": per, "total": (tot1 / totN) if totN > 0 else None,
                                             "single_gpu_stage_ms": {k: single["stage_ms"][k] for k in sh},
                                             "sharded_stage_ms_max_over_ranks": worst,
                                             "note": "draw_f, draw_fstar, draw_beta of the FULL problem on rank 0's GPU alone / the same "
                                                     "stages of the sharded run (slowest rank); draw_beta runs beside the factorisation's "
                                                     "last outer panel on both sides and is counted where it shows (None: < 1 us)"}
                out["whole_iteration_speedup"] = out["value"] / single["iterations_per_s"]
                out["single_gpu_iterations_per_s"] = single["iterations_per_s"]
            else:
                out["item_shard_speedup"] = None
                out["whole_iteration_speedup"] = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, m, ss.engine, y)
            except Exception as e:      # the baseline is a reported extra; never lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
