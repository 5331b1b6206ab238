R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tf_prof -- python3 $R/tools/_tf_time.py > $R/gpurun_out/tf_prof.log 2>&1
cd $R; grep "mode" gpurun_out/tf_prof.log
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/tf_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"])
PY
