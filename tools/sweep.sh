#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/sweep.sh <mode> "<env A>" "<env B>" ...'): one probe under a list of environment
# settings (the library reads GPIRT_* once per process), the default re-run between them so that drift of the box shows.
#   mode iter    tools/iter_time.py: whole iteration, min / mean of 4 x 10 steps
#        factor  tools/factor_time.py 8192 15: the factorisation alone
#        bench   bench.py --steps 20: rate, factor stage, roofline fraction by class
# (replaces round 2/3's sweep_env / sweep_stages / sweep_iter* / ab_iter / sweep_nbp / gap_sweep scripts)
cd ${GRAFT_REPO_ROOT:-.}
mode=$1; shift
run() {
  case $mode in
    iter)   env $1 timeout -k 10 120 python tools/iter_time.py 4 10 2>&1 | grep lowrank | cut -c1-62 ;;
    factor) env $1 timeout -k 10 120 python tools/factor_time.py 8192 15 2>&1 | tail -1 ;;
    bench)  env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt-forms --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), d['config']['stage_ms'], round(d['roofline']['frac'],3), {k: (round(v['frac'],3) if v['frac'] else None, round(v['ms_per_step'],2)) for k,v in d['roofline']['by_class'].items()})" ;;
    *) echo "mode: iter | factor | bench"; exit 2 ;;
  esac
}
for cfg in "$@"; do
  echo "default        $(run A=0)"
  echo "$cfg   $(run "$cfg")"
done
