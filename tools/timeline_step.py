"""Kernel timeline of the LAST MCMC iteration of tools/step_only.py under rocprofv3 --kernel-trace: every launch in order with its
duration, grouped by stage marker kernels.  usage: python tools/timeline_step.py <trace dir>"""
import csv, glob, os, sys, collections
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last iteration starts at the last item_fill_kernel launch that precedes a trmm (draw_f)
starts = [i for i, r in enumerate(rows) if "item_fill" in r["Kernel_Name"]]
# iterations issue item_fill once in draw_f (z) -- take the last but keep everything after it
idx = starts[-1]
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
agg = collections.OrderedDict()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("gpirt::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    key = nm
    if key not in agg: agg[key] = [0, 0.0, (s - t0) / 1e3]
    agg[key][0] += 1; agg[key][1] += (e - s) / 1e3
t1 = max(int(r["End_Timestamp"]) for r in rows)
print(f"iteration span {(t1 - t0) / 1e3:.1f} us, {len(rows)} launches")
for k, (c, t, first) in agg.items():
    print(f"  first@{first:8.1f}  {t:8.1f} us  {c:4d} x {t / c:7.1f}  {k}")
