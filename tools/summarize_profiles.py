"""Turn gpurun_out/profiles_raw (tools/collect_profiles.sh) into the committed profiles/ summaries.
usage: python tools/summarize_profiles.py [raw dir] [tag]"""
import collections, csv, glob, json, os, shutil, sys

raw = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profiles_raw"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
SYRK = {"gemm_f64_kernel<false, true, 128, 8": "128-tile", "gemm_f64_kernel<false, true, 64, 8": "64-tile"}
PANEL = "panel_ll_kernel"


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)   # gpurun merges old calls' pid-named files too


def short(name):
    return name.replace("gpirt::(anonymous namespace)::", "").replace("gpirt::", "").replace("void ", "")


stats = newest(raw + "/stats/**/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
bench = json.loads([l for l in open(raw + "/stats.log") if l.startswith("{")][-1])


def pmc(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(f"{raw}/{name}/**/*counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        for key in list(SYRK) + [PANEL]:
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in agg.items()}


fe, wr, mf = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_mfma")
rl = bench["roofline"]
cls_of = {"128-tile": ["trailing_128tile"], "64-tile": ["trailing_64tile", "in_panel_k512"]}
per_kernel = {}
tot_traffic = tot_launch = 0.0
for key, label in SYRK.items():
    if key not in fe:
        continue
    fetch_kb, nl = fe[key]["FETCH_SIZE"]
    write_kb = wr[key]["WRITE_SIZE"][0]
    traffic = (2 * fetch_kb + write_kb) * 1024          # FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B
    gui, busy = mf[key]["GRBM_GUI_ACTIVE"][0], mf[key]["SQ_VALU_MFMA_BUSY_CYCLES"][0]
    cl = [rl["by_class"][c] for c in cls_of[label] if rl["by_class"][c]["launches"]]
    alg = (sum(c["algorithmic_bytes_per_launch"] * c["launches"] for c in cl) / sum(c["launches"] for c in cl)) if cl else None
    per_kernel[label] = {"kernel": key + ", false>", "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
                         "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
                         "traffic_over_algorithmic": (traffic / alg) if alg else None,
                         "mfma_busy_frac": busy / ((gui / 8) * 1024), "launches_sampled": nl}
    st = [r for r in rows if key in r["Name"]]
    n_l = float(st[0]["Calls"]) if st else nl
    tot_traffic += traffic * n_l
    tot_launch += n_l
json.dump({"kernel": "gemm_f64_kernel<false, true, T, 8, false>, T = 128 and 64 (every syrk launch of the factorisation)",
           "hbm_bytes_per_launch": tot_traffic / tot_launch if tot_launch else None,
           "by_kernel": per_kernel,
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 1`, "
                     f"round {tag[1:]}; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md "
                     f"HBM section); launch-weighted mean over both tile instantiations; Infinity-Cache hits are counted",
           }, open("profiles/trailing_traffic.json", "w"), indent=1)

with open(f"profiles/{tag}_summary.md", "w") as f:
    f.write(f"# Round {tag[1:]} profiles (MI355X, ROCm 7.2, `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-forms` under rocprofv3)\n\n")
    f.write(f"Raw per-kernel statistics: `profiles/{tag}_kernel_stats.csv` (rocprofv3 --kernel-trace --stats); collected by "
            f"`tools/collect_profiles.sh`, summarised by `tools/summarize_profiles.py`.\n\n")
    f.write(f"bench.py line of the profiled run: {bench['value']:.2f} it/s, {bench['ms_per_step']:.2f} ms/step, stages {bench['config']['stage_ms']}\n\n")
    f.write("## The factorisation's syrk launches (`roofline` of bench.py): fp64 MFMA, lower trapezoid\n\n")
    f.write("| kernel | source | launches | avg duration (ms) | achieved TFLOP/s | frac of 78.6 |\n|---|---|---|---|---|---|\n")
    for key, label in SYRK.items():
        cl = [rl["by_class"][c] for c in cls_of[label] if rl["by_class"][c]["launches"]]
        if cl:
            n_l = sum(c["launches"] for c in cl)
            ms = sum(c["avg_launch_ms"] * c["launches"] for c in cl) / n_l
            fl = sum(c["flops_per_launch"] * c["launches"] for c in cl) / n_l
            f.write(f"| `{key}, false>` | bench.py HIP events (timed region) | {n_l} | {ms:.4f} | {fl / ms / 1e9:.2f} | {fl / ms / 1e9 / 78.6:.3f} |\n")
            st = [r for r in rows if key in r["Name"]]
            if st:
                avg_ms = float(st[0]["AverageNs"]) / 1e6
                f.write(f"| | rocprofv3 kernel stats | {st[0]['Calls']} | {avg_ms:.4f} | {fl / avg_ms / 1e9:.2f} | {fl / avg_ms / 1e9 / 78.6:.3f} |\n")
    f.write(f"\nAll syrk launches together (bench.py `roofline`): {rl['achieved']:.2f} TFLOP/s = {rl['frac']:.3f} of 78.6.\n\n")
    f.write("PMC passes (separate runs, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`), averages per launch:\n\n")
    f.write("| kernel | FETCH_SIZE raw KB | x2 MB | WRITE_SIZE MB | traffic MB | algorithmic MB | ratio | MFMA busy |\n|---|---|---|---|---|---|---|---|\n")
    for label, d in per_kernel.items():
        alg = d["algorithmic_bytes_per_launch"]
        f.write(f"| {label} | {d['fetch_size_kb_raw']:.0f} | {2 * d['fetch_size_kb_raw'] * 1024 / 1e6:.1f} | {d['write_size_kb'] * 1024 / 1e6:.1f} | "
                f"{d['hbm_bytes_per_launch'] / 1e6:.1f} | {alg / 1e6 if alg else float('nan'):.1f} | "
                f"{d['traffic_over_algorithmic'] if d['traffic_over_algorithmic'] else float('nan'):.2f} | {d['mfma_busy_frac']:.3f} |\n")
    f.write("\n(algorithmic = C trapezoid read + written once + the M x K panel operand read once; FETCH_SIZE counts Infinity-Cache hits, "
            "so `traffic` is an upper bound on HBM bytes.)\n\n")
    if PANEL in fe:
        f.write(f"`panel_ll_kernel`: FETCH_SIZE {fe[PANEL]['FETCH_SIZE'][0]:.0f} KB raw, WRITE_SIZE {wr[PANEL]['WRITE_SIZE'][0]:.0f} KB per launch; "
                f"MFMA busy {mf[PANEL]['SQ_VALU_MFMA_BUSY_CYCLES'][0] / ((mf[PANEL]['GRBM_GUI_ACTIVE'][0] / 8) * 1024):.3f} of the chip's SIMDs (latency-bound pivot chain).\n\n")
    f.write("## All kernels (rocprofv3 --stats)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:26]:
        f.write(f"| `{short(r['Name'])[:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
print(open(f"profiles/{tag}_summary.md").read()[:3000])
