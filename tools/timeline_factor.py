"""Per-launch timeline of the LAST factorisation in a rocprofv3 --kernel-trace of tools/factor_only.py: start offset,
duration, queue, grid, kernel -- the side-stream chain (panel kernels + in-panel updates) against the main-stream updates.
usage: python tools/timeline_factor.py <trace dir>"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "se_kernel" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
qkey = "Queue_Id" if "Queue_Id" in rows[0] else None
skey = "Stream_Id" if "Stream_Id" in rows[0] else None
print("columns:", list(rows[0].keys()))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("gpirt::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    nm = nm.split("(")[0][:48]
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  q={r.get(qkey, '?') if qkey else '?'} s={r.get(skey, '?') if skey else '?'}  wg={g:6d}  {nm}")
print(f"span {(max(int(r['End_Timestamp']) for r in rows) - t0) / 1e3:.1f} us")
