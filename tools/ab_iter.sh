# usage: bash tools/ab_iter.sh "<env A>" "<env B>" [rounds]  -- alternates the two settings, prints min/mean per run
A="$1"; B="$2"; R=${3:-3}
for i in $(seq 1 $R); do
  env $A timeout -k 10 120 python tools/iter_time.py 5 10 2>&1 | grep lowrank | cut -c1-62 | sed "s/^/A /"
  env $B timeout -k 10 120 python tools/iter_time.py 5 10 2>&1 | grep lowrank | cut -c1-62 | sed "s/^/B /"
done
