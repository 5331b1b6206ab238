#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel trace of tools/factor_only.py -> per-launch timeline of the last factorisation.
# usage: bash tools/trace_factor.sh <tag> [env assignments...]
TAG=${1:-tl}; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/tl -- python3 $R/tools/factor_only.py $FACTOR_ARGS > $R/gpurun_out/$TAG/tl.log 2>&1
cd $R && python tools/timeline_factor.py gpurun_out/$TAG/tl > gpurun_out/$TAG/timeline.txt 2>&1
tail -2 gpurun_out/$TAG/timeline.txt
