"""Turn gpurun_out/profiles_raw (tools/collect_profiles.sh) into the committed profiles/ summaries.
usage: python tools/summarize_profiles.py [raw dir] [tag]"""
import collections, csv, glob, json, os, shutil, sys

raw = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profiles_raw"
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
SYRK = {"gemm_f64_kernel<false, true, 128, 8": "128-tile", "gemm_f64_kernel<false, true, 64, 8": "64-tile",
        "gemm_f64_kernel<false, true, 64, 16": "in-panel 64-tile"}
PANEL = "panel_ll_kernel"
TF = "tf_mfma_kernel"


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
        for key in list(SYRK) + [PANEL, TF]:
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in agg.items()}


fe, wr, mf = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_mfma")
rl = bench["roofline"]
cls_of = {"128-tile": ["trailing_128tile"], "64-tile": ["trailing_64tile"], "in-panel 64-tile": ["in_panel_update"]}
commit = open(raw + "/commit.txt").read().strip() if os.path.exists(raw + "/commit.txt") else "unknown"
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
json.dump({"kernel": "gemm_f64_kernel<false, true, T, PAD, false>: trailing / deferred updates (PAD = 8, T = 128 and 64) and in-panel updates (PAD = 16) of the factorisation",
           "library_commit": commit,
           "hbm_bytes_per_launch": tot_traffic / tot_launch if tot_launch else None,
           "by_kernel": per_kernel,
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 1`, "
                     f"round {tag[1:]}, library commit {commit}; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md "
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
    if TF in fe and TF in wr and TF in mf:
        e = rl.get("theta_int8_product") or {}
        tf_traffic = (2 * fe[TF]["FETCH_SIZE"][0] + wr[TF]["WRITE_SIZE"][0]) * 1024
        alg = e.get("algorithmic_bytes_per_launch")
        f.write(f"`tf_mfma_kernel` (draw_theta's log-posterior product in fixed point, int8 MFMA): FETCH_SIZE {fe[TF]['FETCH_SIZE'][0]:.0f} KB raw "
                f"(x2 = {2 * fe[TF]['FETCH_SIZE'][0] * 1024 / 1e6:.1f} MB), WRITE_SIZE {wr[TF]['WRITE_SIZE'][0] * 1024 / 1e6:.1f} MB per launch = "
                f"{tf_traffic / 1e6:.1f} MB of traffic" + (f" against {alg / 1e6:.1f} MB algorithmic (both operands once + the fp64 result: {tf_traffic / alg:.2f}x; "
                f"the digit planes are re-read by the 32 respondent tiles out of the L2s / the Infinity Cache, which FETCH_SIZE counts)" if alg else "") +
                f"; MFMA busy {mf[TF]['SQ_VALU_MFMA_BUSY_CYCLES'][0] / ((mf[TF]['GRBM_GUI_ACTIVE'][0] / 8) * 1024):.3f} of the chip's SIMDs.\n\n")
    if PANEL in fe:
        f.write(f"`panel_ll_kernel`: FETCH_SIZE {fe[PANEL]['FETCH_SIZE'][0]:.0f} KB raw, WRITE_SIZE {wr[PANEL]['WRITE_SIZE'][0]:.0f} KB per launch; "
                f"MFMA busy {mf[PANEL]['SQ_VALU_MFMA_BUSY_CYCLES'][0] / ((mf[PANEL]['GRBM_GUI_ACTIVE'][0] / 8) * 1024):.3f} of the chip's SIMDs (latency-bound pivot chain).\n\n")
    f.write("## All kernels (rocprofv3 --stats)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:26]:
        f.write(f"| `{short(r['Name'])[:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")


def stat_row(rows_, key):
    st = [r for r in rows_ if key in r["Name"]]
    return st[0] if st else None


with open(f"profiles/{tag}_summary.md", "a") as f:
    f.write(f"\n## The other roofline entries of bench.py's line, against rocprofv3's averages (library commit {commit})\n\n")
    f.write("| entry | kernel | bench.py: launches, avg ms, frac | rocprofv3: calls, avg ms | frac from the rocprofv3 average |\n|---|---|---|---|---|\n")
    bench_ref = None
    if os.path.exists(raw + "/stats_ref.log"):
        lines = [l for l in open(raw + "/stats_ref.log") if l.startswith("{")]
        bench_ref = json.loads(lines[-1]) if lines else None
    for name, key in (("draw_f_trmm", "gemm_f64_kernel<false, false, 128, 0"), ("theta_int8_product", "tf_mfma_kernel"),
                      ("replay_products", "rs3p_products_kernel")):
        if name == "replay_products":
            # from the run WITH the default-contract leg; rocprofv3's average over the REAL passes of that run (spare passes,
            # which find every item done and leave at once, are not passes over L -- bench.py's events leave them out too)
            if not bench_ref:
                continue
            e = bench_ref["roofline"].get(name)
            tr = newest(raw + "/stats_ref/**/*kernel_trace.csv")
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(tr)) if key in r["Kernel_Name"]]
            real = [d for d in dur if d > 0.25 * max(dur)]
            st = {"Calls": f"{len(real)} real of {len(dur)}", "AverageNs": 1e6 * sum(real) / len(real)} if real else None
        else:
            e = rl.get(name)
            st = stat_row(rows, key)
        if not e or not st:
            continue
        avg_ms = float(st["AverageNs"]) / 1e6
        work = e["flops_per_launch"] if e["bound"] == "mfma" else e["algorithmic_bytes_per_launch"]
        frac = work / (avg_ms * 1e-3) / (1e12 if e["bound"] == "mfma" else 1e9) / e["peak"]
        f.write(f"| `{name}` ({e['bound']}) | `{key}...` | {e['launches']}, {e['avg_launch_ms']:.4f}, {e['frac']:.3f} | {st['Calls']}, {avg_ms:.4f} | {frac:.3f} |\n")
    fo = rl.get("factor_overall")
    if fo and fo.get("frac"):
        f.write(f"\n`factor_overall`: n^3 / 3 = {fo['flops']:.3e} flop in the factor stage's {fo['stage_ms']:.3f} ms = {fo['achieved']:.1f} TFLOP/s = "
                f"{fo['frac']:.3f} of 78.6 (pivot chain, sub-panel kernels and all).\n")
    rr = (bench_ref or bench)["config"].get("reference_rng")
    if rr:
        f.write(f"\nDefault contract in the same run (`config.reference_rng`): {rr.get('value')} iterations/s, "
                f"{rr.get('passes_over_L_per_iteration')} passes over L per iteration ({rr.get('items_per_pass')} items per pass).\n")

# ---- the default contract (R-stream replay): tools/rstream_step.py 8192 1024 under rocprofv3, with the structured pass of the
# predictor (the default from 4096 respondents on, csrc/rs_lr.hip) and with the dense single-precision pass (GPIRT_RS_LR=2)
def replay_section(sub, structured):
    try:
        rstats = newest(f"{raw}/{sub}_stats/**/*kernel_stats.csv")
    except ValueError:
        return
    name = "replay" if structured else "replay_dense"
    shutil.copy(rstats, f"profiles/{tag}_{name}_kernel_stats.csv")
    rrows = list(csv.DictReader(open(rstats)))
    log = [l.strip() for l in open(f"{raw}/{sub}_stats.log") if ("per iteration" in l or "stage ms" in l or "rejection counts" in l)]

    def one(pm, counter):
        try:
            path = newest(f"{raw}/{sub}_{pm}/**/*counter_collection.csv")
        except ValueError:
            return (None, 0)
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if "rs3p_products" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        big = [v for v in vals if v > 0.5 * max(vals)] if vals else []       # (spare passes leave at once: not a pass)
        return (sum(big) / len(big), len(big)) if big else (None, 0)

    fk, nf = one("pmc_fetch", "FETCH_SIZE")
    wk, _ = one("pmc_write", "WRITE_SIZE")
    busy, _ = one("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
    gui, _ = one("pmc_mfma", "GRBM_GUI_ACTIVE")
    n_ = 8192
    alg = 4.0 * (512 + 64) * n_ if structured else 4.0 * n_ * (n_ + 1) / 2
    with open(f"profiles/{tag}_{name}_summary.md", "w") as f:
        f.write(f"# Round {tag[1:]}: the default contract (R-stream replay), " + ("" if structured else "`GPIRT_RS_LR=2` (the predictor's DENSE pass), ") +
                f"`rocprofv3 --kernel-trace --stats -- python3 tools/rstream_step.py 8192 1024` (library commit {commit})\n\n")
        f.write("Init + five iterations at 8192 x 1024.  draw_f is PREDICT + VERIFY (csrc/rs_predict.hip, DESIGN.md section 2): the starts of all items "
                "in R's stream are predicted by a chain of passes, each placing up to four items from 32 candidate starts (1 + 13 + 11 + 7) -- ")
        if structured:
            f.write("`rs3p_products_kernel` = the STRUCTURED pass (csrc/rs_lr.hip): L32 z over the 512-column part that holds each row group's diagonal, "
                    "plus y_J = C_J z_J for the parts' coefficient blocks (the blocks of L below the diagonal parts are V C: the Lagrange basis of 64 "
                    "Chebyshev nodes at theta times coefficients built once per iteration from theta and the factor's 64 x 64 diagonal blocks by "
                    "`lr_basis_kernel`, `lr_gram_kernel`, `lr_scan_kernel`, `lr_coef_kernel`); `rs_lr_apply_kernel` = prefix over the parts' records + V x prefix; ")
        else:
            f.write("`rs3p_products_kernel` = L32 z for the pass's candidates over the whole lower triangle as single-precision tiles; ")
        f.write("`rs3p_decide_kernel` = the first 16 trial points of every candidate side by side, one ticket, the "
                "last work-group decides the four slots -- then `rs_gather_kernel` + ONE triangular fp64 product (`gemm_f64_kernel<false, false, 128, ...>`) "
                "+ `rs_verify_kernel` (every slice loop exactly, side by side) + `rs_commit_*` (accept in order up to the first misprediction).  "
                "`rs3_begin_kernel` = the normal that starts at every position of the iteration's window; `rs32_tile_kernel` = L re-tiled "
                "once per iteration as floats; `rs_unpack_kernel` = Mersenne-Twister words -> unif_rand() values.\n\n")
        for l in log:
            f.write(l + "\n\n")
        f.write("(Under rocprofv3 a launch costs the host more -- a draw is ~700-1000 launches -- so the iteration above reads a few ms longer than unprofiled "
                "(`profiles/r06_bench.json`: `config.reference_rng`; `profiles/r06_predictor_stats.txt` for long chains).  Per-kernel durations are unaffected.)\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rrows[:16]:
            f.write(f"| `{short(r['Name'])[:80]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
        tr = newest(f"{raw}/{sub}_stats/**/*kernel_trace.csv")
        trows = list(csv.DictReader(open(tr)))

        def real_avg(kname):
            # the spare passes (every item already predicted) leave at once: the average over REAL passes is what counts
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trows if kname in r["Kernel_Name"]]
            real = [d for d in dur if d > 0.4 * sorted(dur)[(len(dur) * 3) // 4]] if dur else []
            return (sum(real) / len(real), len(real), len(dur)) if real else (None, 0, 0)

        avg_us, nreal, nall = real_avg("rs3p_products")
        if avg_us:
            if structured:
                f.write(f"\n`rs3p_products_kernel` (structured): {nreal} real passes of {nall} launches (the others find every item predicted and leave at once), "
                        f"{avg_us:.1f} us per real pass for 4 (512 + 64) n = {alg / 1e6:.1f} MB (the dense pass: {4.0 * n_ * (n_ + 1) / 2 / 1e6:.0f} MB) -> "
                        f"{alg / avg_us / 1e6:.2f} TB/s = {alg / avg_us / 1e6 / 8.0:.3f} of the 8 TB/s HBM peak: the kernel is bound by its launch and two "
                        f"dependent memory round trips (anchor, then windows and tiles), not by bytes.\n")
                a_us, na, _ = real_avg("rs_lr_apply_kernel")
                if a_us:
                    f.write(f"\n`rs_lr_apply_kernel`: {a_us:.1f} us per real pass.\n")
            else:
                f.write(f"\n`rs3p_products_kernel`: {nreal} real passes of {nall} launches (the others find every item predicted and leave at once), "
                        f"{avg_us:.1f} us per real pass -> {alg / avg_us / 1e6:.2f} TB/s of L as floats (algorithmic 4 n (n + 1) / 2 = {alg / 1e6:.0f} MB) = "
                        f"{alg / avg_us / 1e6 / 8.0:.3f} of the 8 TB/s HBM peak; its 2 x 32 x n (n + 1) / 2 = {64 * n_ * (n_ + 1) / 2 / 1e9:.2f} GFLOP per pass = "
                        f"{64 * n_ * (n_ + 1) / 2 / avg_us / 1e6:.1f} TFLOP/s = {64 * n_ * (n_ + 1) / 2 / avg_us / 1e6 / 157.3:.3f} of the 157.3 TFLOP/s f32-input MFMA peak.\n")
            d_us, nd, ndall = real_avg("rs3p_decide_kernel")
            if d_us:
                f.write(f"\n`rs3p_decide_kernel`: {d_us:.1f} us per real pass ({nd} of {ndall} launches).\n")
        if fk:
            f.write(f"\nPMC passes of `tools/rstream_step.py 8192 128` (separate runs), averages over the {nf} real passes: FETCH_SIZE {fk:.0f} KB raw "
                    f"(x 2 = {2 * fk * 1024 / 1e6:.0f} MB: gfx950 tallies 128-B requests at 64 B), WRITE_SIZE {wk:.0f} KB ({wk * 1024 / 1e6:.1f} MB: the parts of the "
                    f"32 products) -> traffic {(2 * fk + wk) * 1024 / 1e6:.0f} MB = {(2 * fk + wk) * 1024 / alg:.2f} x algorithmic")
            if busy and gui:
                f.write(f"; MFMA busy {busy / ((gui / 8) * 1024):.3f} of the chip's SIMDs")
            f.write(".\n")
    print(open(f"profiles/{tag}_{name}_summary.md").read()[:2500])


replay_section("replay", True)
replay_section("replay_dense", False)
print(open(f"profiles/{tag}_summary.md").read()[:3000])
