#!/bin/bash
# usage: gpurun -- 'bash tools/sweep_nbp.sh'   -- sub-panel width of the persistent panel kernel, whole iteration
mkdir -p gpurun_out/nbp
for w in 512 384 448 576 640 512; do
  echo "== GPIRT_NBP=$w"
  GPIRT_NBP=$w timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt-forms > gpurun_out/nbp/b.log 2>&1 || { tail -5 gpurun_out/nbp/b.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/nbp/b.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["config"]["stage_ms"]["factor"])
PY
done
