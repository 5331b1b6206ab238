#!/bin/bash
# Run ON THE GPU BOX: tools/iter_time.py (whole iteration, min / mean of 4 x 10 steps) under a list of environment settings,
# the default ("A=0") re-run between them so that drift of the box shows:   bash tools/sweep_iter2.sh "GPIRT_NBP=384" ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env A=0 timeout -k 10 120 python tools/iter_time.py 4 10 2>&1 | grep lowrank | cut -c1-58 | sed "s/^/default       /"
  env $cfg timeout -k 10 120 python tools/iter_time.py 4 10 2>&1 | grep lowrank | cut -c1-58 | sed "s/^/$cfg  /"
done
