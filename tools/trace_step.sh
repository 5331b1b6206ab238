#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel trace of tools/step_only.py -> timelines of the last iteration.
# usage: bash tools/trace_step.sh <tag> [env assignments...]
TAG=${1:-ts}; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/tl -- python3 $R/tools/step_only.py > $R/gpurun_out/$TAG/tl.log 2>&1
cd $R && python tools/timeline_iter.py gpurun_out/$TAG/tl > gpurun_out/$TAG/timeline_iter.txt 2>&1
python tools/timeline_step.py gpurun_out/$TAG/tl > gpurun_out/$TAG/timeline_step.txt 2>&1
tail -3 gpurun_out/$TAG/timeline_iter.txt
