"""One-GPU emulation of the per-rank iteration of an item-sharded run at the metric size: bench.py at m / G items for G = 1, 2, 4, 8
(what ONE rank of G does, minus the collectives), headline preset and the as-written draw_fstar; prints the prediction table of
DESIGN.md section 7 (per-stage ms, implied item-shard speed-up per stage, whole-iteration figure).  Run on the GPU box:
    python tools/shard_emulation.py [n = 8192] [m = 1024] > profiles/r06_multigpu_prediction.md"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rows = {}
for form in ("lowrank", "double_solve"):
    for G in (1, 2, 4, 8):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", str(n), "--m", str(m // G), "--steps", "20", "--warmup", "5",
                              "--fstar", form, "--no-cpu-baseline", "--no-reference-rng", "--no-alt-forms"], capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            print("FAILED", form, G, out.stderr[-500:], file=sys.stderr); continue
        d = json.loads(line[-1])
        rows[(form, G)] = (d["ms_per_step"], d["config"]["stage_ms"])
        print(f"[{form} m/G={m // G}] {d['ms_per_step']:.3f} ms {d['config']['stage_ms']}", file=sys.stderr, flush=True)
stages = ("factor", "draw_f", "draw_fstar", "theta_gemm", "theta_sample", "draw_beta")
print(f"# One-GPU emulation of the per-rank iteration at N = {n} x m = {m} (round 6): `python tools/shard_emulation.py {n} {m}`\n")
print("What ONE rank of G GPUs executes per iteration with the factorisation replicated (`--chol replicated`), minus the collectives "
      "(the all-gather of f*: N x m doubles = %.1f MB, and the %d-byte all-reduce of theta): `bench.py --m %d/G`, 20 timed steps, "
      "stage times from device events of two extra iterations.  draw_theta is respondent-sharded in the real run (`theta=gather`: "
      "each rank forms the log-posterior of n/G respondents over ALL m items), which this emulation cannot show -- its entries here "
      "are the all-respondents x m/G-items product, an upper bound of the same order.\n" % (1001 * m * 8 / 1e6, 8 * n, m))
for form, title in (("lowrank", "`gpirt_fast_options()` (the headline preset: fused + rank-64 draw_fstar)"), ("double_solve", "draw_fstar as written (`src/draw-fstar.cpp:17-25`), item-keyed RNG")):
    if (form, 1) not in rows:
        continue
    print(f"## {title}\n")
    print("| G | items per rank | ms per iteration | it/s ceiling before communication | whole-iteration speed-up | " + " | ".join(stages) + " |")
    print("|---|---|---|---|---|" + "---|" * len(stages))
    base_ms, base_st = rows[(form, 1)]
    for G in (1, 2, 4, 8):
        if (form, G) not in rows:
            continue
        ms, st = rows[(form, G)]
        print(f"| {G} | {m // G} | {ms:.3f} | {1e3 / ms:.1f} | {base_ms / ms:.2f}x | " + " | ".join(f"{st.get(k, 0):.3f}" for k in stages) + " |")
    print("\nImplied item-shard speed-up of the item-sharded stages (stage time at G = 1 / stage time at m / G items):\n")
    print("| G | draw_f | draw_fstar | draw_f + draw_fstar + draw_beta |")
    print("|---|---|---|---|")
    for G in (2, 4, 8):
        if (form, G) not in rows:
            continue
        ms, st = rows[(form, G)]
        sh = ("draw_f", "draw_fstar", "draw_beta")
        t1 = sum(base_st.get(k, 0) for k in sh); tG = sum(st.get(k, 0) for k in sh)
        def sp(k):
            return f"{base_st[k] / st[k]:.2f}x" if st.get(k, 0) > 1e-3 else "-"
        print(f"| {G} | {sp('draw_f')} | {sp('draw_fstar')} | {t1 / tG:.2f}x |")
    print()
