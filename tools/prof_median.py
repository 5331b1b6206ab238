"""Median / p10 / p90 duration per kernel of a rocprofv3 --kernel-trace directory (spare launches that leave at once distort the averages).
usage: python tools/prof_median.py DIR [substring ...]"""
import collections, csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pats = sys.argv[2:]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("gpirt::(anonymous namespace)::", "").replace("void ", "")[:70]
    if not pats or any(p in k for p in pats):
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v = np.array(v)
    print(f"{k:70s} n={len(v):5d} median {np.median(v):8.1f}  p10 {np.quantile(v, .1):8.1f}  p90 {np.quantile(v, .9):8.1f} us  total {v.sum() / 1e3:8.2f} ms")
