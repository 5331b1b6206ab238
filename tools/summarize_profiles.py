"""Turn gpurun_out/profiles_raw (tools/collect_profiles.sh) into the committed profiles/ summaries."""
import collections, csv, glob, json, os, shutil, sys
raw = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profiles_raw"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
KERNEL = "gemm_f64_kernel<false, true, 128, 8"
stats = max(glob.glob(raw + "/stats/runc/*kernel_stats.csv"), key=os.path.getmtime)   # gpurun merges old calls' pid-named files too
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
bench = json.loads([l for l in open(raw + "/stats.log") if l.startswith("{")][-1])

def pmc(name):
    f = max(glob.glob(f"{raw}/{name}/runc/*counter_collection.csv"), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

fe, wr, mf = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_mfma")
fetch_kb, write_kb = fe["FETCH_SIZE"][0], wr["WRITE_SIZE"][0]
traffic = (2 * fetch_kb + write_kb) * 1024
gui, busy = mf["GRBM_GUI_ACTIVE"][0], mf["SQ_VALU_MFMA_BUSY_CYCLES"][0]
util = busy / ((gui / 8) * 1024)
tr = [r for r in rows if KERNEL in r["Name"]][0]
json.dump({"kernel": KERNEL + ", false> (potrf trailing update)", "hbm_bytes_per_launch": traffic,
           "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 reports "
                   "half of wide coalesced reads, MI355X_MICROARCH.md HBM section); average over the 128-tile trailing "
                   "launches of bench.py",
           "mfma_util_pmc": util, "launches_sampled": fe["FETCH_SIZE"][1]},
          open("profiles/trailing_traffic.json", "w"), indent=1)
rl = bench["roofline"]
with open(f"profiles/{tag}_summary.md", "w") as f:
    f.write(f"# Round {tag[1:]} profiles (MI355X, ROCm 7.2, `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-forms` under rocprofv3)\n\n")
    f.write(f"Raw per-kernel statistics: `profiles/{tag}_kernel_stats.csv` (rocprofv3 --kernel-trace --stats); collected by `tools/collect_profiles.sh`, summarised by `tools/summarize_profiles.py`.\n\n")
    f.write(f"bench.py line of the profiled run: {bench['value']:.2f} it/s, {bench['ms_per_step']:.2f} ms/step, stages {bench['config']['stage_ms']}\n\n")
    f.write(f"## Dominant kernel: potrf trailing update `{KERNEL}, false>` (fp64 MFMA syrk, lower blocks)\n\n")
    f.write("| source | launches | avg duration (ms) | achieved TFLOP/s | frac of 78.6 |\n|---|---|---|---|---|\n")
    f.write(f"| bench.py HIP events (timed region) | {rl['launches']} | {rl['avg_launch_ms']:.4f} | {rl['achieved']:.2f} | {rl['frac']:.3f} |\n")
    avg_ms = float(tr["AverageNs"]) / 1e6
    f.write(f"| rocprofv3 kernel stats | {tr['Calls']} | {avg_ms:.4f} | {rl['flops_per_launch'] / avg_ms / 1e9:.2f} | {rl['flops_per_launch'] / avg_ms / 1e9 / 78.6:.3f} |\n\n")
    f.write("PMC passes (separate runs, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`), averages per launch of that kernel:\n\n")
    f.write(f"* FETCH_SIZE {fetch_kb:.0f} KB raw (x2 gfx950 correction = {2 * fetch_kb * 1024 / 1e6:.1f} MB), WRITE_SIZE {write_kb:.0f} KB ({write_kb * 1024 / 1e6:.1f} MB) -> HBM traffic {traffic / 1e6:.1f} MB per launch\n")
    f.write(f"* SQ_VALU_MFMA_BUSY_CYCLES {busy:.3e}, GRBM_GUI_ACTIVE {gui:.3e} (sum over 8 XCDs) -> MFMA busy fraction {util:.3f} of the 1024 SIMDs while the kernel runs\n\n")
    f.write("## All kernels (rocprofv3 --stats)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:24]:
        nm = r["Name"].replace("gpirt::(anonymous namespace)::", "").replace("void ", "")[:70]
        f.write(f"| `{nm}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
print(open(f"profiles/{tag}_summary.md").read()[:1800])
