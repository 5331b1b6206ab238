"""Turn gpurun_out/cfg_TAG (tools/collect_config.sh) into profiles/<round>_<tag>_summary.md + _kernel_stats.csv.
usage: python tools/config_summary.py gpurun_out/cfg_c4 r06_c4 "BASELINE config C4 (N = 8192 x m = 2048, one MI355X)" """
import csv, glob, json, os, shutil, sys

raw, tag, title = sys.argv[1], sys.argv[2], sys.argv[3]
PEAK = 78.6


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def short(name):
    return name.replace("gpirt::(anonymous namespace)::", "").replace("gpirt::", "").replace("void ", "")


b = json.load(open(raw + "/bench.json"))
lines = [l for l in open(raw + "/stats.log") if l.startswith("{")]
bp = json.loads(lines[-1])
stats = newest(raw + "/stats/**/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
cmd = open(raw + "/cmd.txt").read().strip() if os.path.exists(raw + "/cmd.txt") else ""
SYM = {"trailing_128tile": "gemm_f64_kernel<false, true, 128, 8", "trailing_64tile": "gemm_f64_kernel<false, true, 64, 8",
       "in_panel_update": "gemm_f64_kernel<false, true, 64, 16"}
with open(f"profiles/{tag}_summary.md", "w") as f:
    rl, cfg = b["roofline"], b["config"]
    f.write(f"# {title}: `{cmd or 'bench.py'}`\n\n")
    f.write(f"Unprofiled run: **{b['value']:.1f} iterations/s**, {b['ms_per_step']:.3f} ms per iteration, stages {cfg['stage_ms']}; "
            f"by draw_fstar form {cfg.get('iterations_per_s_by_form')}; the default contract (R-stream replay) "
            f"{cfg.get('reference_rng_iterations_per_s')} it/s ({(cfg.get('reference_rng') or {}).get('iterations_per_s_by_fstar_form')}). "
            f"kernel_fp32 = {cfg.get('kernel_fp32')}; lowrank_check {cfg.get('lowrank_check') and {k: cfg['lowrank_check'][k] for k in ('max_abs_fstar_lowrank_minus_full_solve', 'max_abs_fstar_lowrank_minus_as_written', 'max_abs_fstar', 'passed')}}.\n\n")
    f.write("Roofline entries of that line (HIP event pairs of the library on the launch's own stream):\n\n")
    f.write("| entry | launches | avg ms | flop (or B) per launch | achieved | frac |\n|---|---|---|---|---|---|\n")
    for k, c in rl["by_class"].items():
        if c["launches"]:
            f.write(f"| syrk `{k}` | {c['launches']} | {c['avg_launch_ms']:.4f} | {c['flops_per_launch']:.4g} | {c['achieved']:.1f} TFLOP/s | {c['frac']:.3f} |\n")
    for k in ("draw_f_trmm", "theta_int8_product", "replay_products"):
        e = rl.get(k)
        if e:
            w = e["flops_per_launch"] if e["bound"] == "mfma" else e["algorithmic_bytes_per_launch"]
            f.write(f"| `{k}` ({e['bound']}) | {e['launches']} | {e['avg_launch_ms']:.4f} | {w:.4g} | {e['achieved']:.1f} {e['unit']} | {e['frac']:.3f} |\n")
    fo = rl["factor_overall"]
    f.write(f"| `factor_overall` | 1 stage | {fo['stage_ms']:.3f} | {fo['flops']:.4g} | {fo['achieved']:.1f} TFLOP/s | {fo['frac']:.3f} |\n\n")
    f.write(f"All syrk launches: {rl['achieved']:.1f} TFLOP/s = {rl['frac']:.3f} of {PEAK}.\n\n")
    f.write(f"Profiled run (`rocprofv3 --kernel-trace --stats`, same command + `--no-alt-forms --no-reference-rng`): {bp['value']:.1f} it/s; "
            f"per-kernel statistics in `profiles/{tag}_kernel_stats.csv`:\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:16]:
        f.write(f"| `{short(r['Name'])[:80]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    f.write("\nRoofline of each syrk symbol recomputed from the rocprofv3 average (flop per launch from the profiled run's own events):\n\n")
    for k, key in SYM.items():
        c = bp["roofline"]["by_class"].get(k)
        st = [r for r in rows if key in r["Name"]]
        if c and c["launches"] and st:
            avg_ms = float(st[0]["AverageNs"]) / 1e6
            fr = c["flops_per_launch"] / avg_ms / 1e9 / PEAK
            f.write(f"* `{k}` (`{key}, false>`): rocprofv3 average {avg_ms * 1e3:.1f} us over {st[0]['Calls']} launches -> "
                    f"{c['flops_per_launch'] / avg_ms / 1e9:.1f} TFLOP/s = **{fr:.3f}** of {PEAK} (events of the same profiled run: "
                    f"{c['avg_launch_ms'] * 1e3:.1f} us, {c['frac']:.3f}; unprofiled line: {rl['by_class'][k]['frac']:.3f}).\n")
    e = bp["roofline"].get("draw_f_trmm")
    st = [r for r in rows if "gemm_f64_kernel<false, false, 128, 0" in r["Name"]]
    if e and st:
        avg_ms = float(st[0]["AverageNs"]) / 1e6
        f.write(f"* `draw_f_trmm`: rocprofv3 average {avg_ms:.3f} ms over {st[0]['Calls']} launches (events: {e['avg_launch_ms']:.3f} ms, {e['frac']:.3f}).\n")
print(open(f"profiles/{tag}_summary.md").read())
