#!/bin/bash
# usage: gpurun -- 'bash tools/cols_ab.sh'   -- GPIRT_PANEL_COLS=1 (progressive hand-off of L_jj) against =0, whole iteration
mkdir -p gpurun_out/cols
for c in 1 0 1 0; do
  echo "== GPIRT_PANEL_COLS=$c"
  GPIRT_PANEL_COLS=$c timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt-forms > gpurun_out/cols/b.log 2>&1 || { tail -5 gpurun_out/cols/b.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/cols/b.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["config"]["stage_ms"])
PY
done
