"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and (optionally) a per-grid breakdown."""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].replace("gpirt::(anonymous namespace)::", "").replace("void ", "")[:60]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[k][0] += 1; agg[k][1] += dur
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:60s} {c:7d} {t/1e3:10.3f} {t/c:9.1f} {100*t/tot:6.2f}")
if len(sys.argv) > 2:
    pat = sys.argv[2]
    g = collections.defaultdict(list)
    for r in rows:
        if pat in r["Kernel_Name"]:
            g[int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("grid-size breakdown for", pat)
    for k in sorted(g):
        v = g[k]; print(f"  blocks={k:6d} calls={len(v):5d} avg_us={sum(v)/len(v):9.1f} total_ms={sum(v)/1e3:9.3f}")
