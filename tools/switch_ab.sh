#!/bin/bash
# Run ON THE GPU BOX (gpurun): the headline iteration under each of the library's numerics-neutral switches, one session.
# usage: bash tools/switch_ab.sh > profiles/rNN_switches_ab.txt
R=${GRAFT_REPO_ROOT:-.}
cd $R
echo "# bench.py --no-reference-rng --no-alt-forms --no-cpu-baseline --steps 60 --warmup 10 (8192 x 1024, gpirt_fast_options), one MI355X, same session"
echo "# switch | iterations/s | ms per iteration | stage ms"
for cfg in "" "GPIRT_THETA_FIXED=2" "GPIRT_ESS_SCREEN=2" "GPIRT_THETA_FIXED=2 GPIRT_ESS_SCREEN=2" "GPIRT_LL_EXACT=1" "GPIRT_EARLY_INV=2" "GPIRT_PREP_EARLY=2" ""; do
  env $cfg python bench.py --no-reference-rng --no-alt-forms --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[${cfg:-default}]', '|', round(d['value'],2), '|', round(d['ms_per_step'],3), '|', d['config']['stage_ms'])"
done
