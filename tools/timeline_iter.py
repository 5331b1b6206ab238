"""Per-launch timeline of the LAST iteration's non-factor part (draw_f .. draw_beta) in a rocprofv3 --kernel-trace of
tools/step_only.py: start offset, duration, queue, grid, kernel -- shows what runs beside what.
usage: python tools/timeline_iter.py <trace dir>"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last iteration: from the last big trmm (gemm_f64_kernel<false, false, 128) to the following se_kernel_lower
trmm = [i for i, r in enumerate(rows) if "gemm_f64_kernel<false, false, 128" in r["Kernel_Name"]]
i0 = trmm[-1]
i0 = max(0, i0 - 3)
i1 = next((i for i in range(trmm[-1], len(rows)) if "se_kernel_lower" in rows[i]["Kernel_Name"]), len(rows) - 1)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("gpirt::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  q={r.get('Queue_Id', '?')}  wg={g:6d}x{r['Grid_Size_Y']:>3s}  {nm}")
