"""Summarise a rocprofv3 kernel trace of tools/factor_only.py: per-kernel busy time and idle gaps of the LAST factorisation.
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [n_factor_runs=3]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factorisation starts at the last se_kernel_lower launch
idx = max(i for i, r in enumerate(rows) if "se_kernel" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = collections.defaultdict(lambda: [0, 0.0])
ivs = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("gpirt::", "").replace("(anonymous namespace)::", "")[:60]
    busy[nm][0] += 1; busy[nm][1] += (e - s) / 1e3
    ivs.append((s, e))
ivs.sort()
cover = 0; cur_s, cur_e = ivs[0]
for s, e in ivs[1:]:
    if s > cur_e: cover += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
cover += cur_e - cur_s
print(f"span {(t1 - t0) / 1e3:.1f} us, GPU busy (union) {cover / 1e3:.1f} us, idle {(t1 - t0 - cover) / 1e3:.1f} us, kernels {len(rows)}")
for k, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t:9.1f} us  {c:5d} x {t / c:8.1f}  {k}")
