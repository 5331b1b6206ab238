cd $GRAFT_REPO_ROOT
for sw in "5 2" "20 5" "40 3" "7 1" "30 10"; do set -- $sw; timeout -k 10 200 python bench.py --no-cpu-baseline --no-alt-forms --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['lowrank_check']; print('steps/warmup $sw', round(d['value'],2), c['max_abs_fstar_lowrank_minus_full_solve'], c['passed'], d['config']['headline_note'])"; done
