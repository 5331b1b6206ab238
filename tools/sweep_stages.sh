#!/bin/bash
# Run ON THE GPU BOX: bench.py stage times under a list of environment settings (one line each)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt-forms --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), d['config']['stage_ms'])"
done
