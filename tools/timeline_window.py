"""Per-launch timeline (start offset, duration, queue, grid, kernel) of ONE iteration in a rocprofv3 --kernel-trace of
tools/step_only.py: from the start of the second-to-last factorisation (se_kernel_lower) to the start of the last one.
usage: python tools/timeline_window.py <trace dir> [min_us = 0: hide launches shorter than this]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
se = [i for i, r in enumerate(rows) if "se_kernel_lower" in r["Kernel_Name"]]
a, b = se[-2], se[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (e - s) / 1e3 < min_us:
        continue
    nm = r["Kernel_Name"].replace("gpirt::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  q={r.get('Queue_Id', '?')}  wg={g:6d}  {nm}")
print(f"iteration (factor start to factor start): {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches")
