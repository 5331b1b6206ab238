#!/bin/bash
# Run ON THE GPU BOX: bench.py under a list of environment settings (one line each), e.g.
#   bash tools/sweep_env.sh "A=1" "GPIRT_DEFER=3" "GPIRT_DEFER=3 GPIRT_PANEL_FIRST=2"
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt-forms --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), d['config']['stage_ms']['factor'], round(d['roofline']['frac'],3), {k: (round(v['frac'],3) if v['frac'] else None, round(v['ms_per_step'],2)) for k,v in d['roofline']['by_class'].items()})"
done
