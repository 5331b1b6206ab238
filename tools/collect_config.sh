#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: one BASELINE config as an unprofiled bench.py line + the same
# command under rocprofv3 --kernel-trace --stats.   usage: tools/collect_config.sh TAG N M [extra bench.py flags]
# Output: gpurun_out/cfg_TAG/{bench.json, bench.log, stats/, stats.log}; tools/config_summary.py turns it into profiles/.
set -e
TAG=$1; N=$2; M=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/cfg_$TAG
mkdir -p $OUT
python3 $R/bench.py --n $N --m $M --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
echo "$TAG bench done: $(python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'])")"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --n $N --m $M --steps 10 --warmup 3 --no-cpu-baseline --no-alt-forms --no-reference-rng "$@" > $OUT/stats.log 2>&1
echo "$TAG stats done"
