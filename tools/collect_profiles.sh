#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats + separate PMC passes of bench.py.
# Output: gpurun_out/profiles_raw/{stats,pmc_fetch,pmc_write,pmc_mfma}/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_raw
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-forms --no-reference-rng > $OUT/stats.log 2>&1
echo "stats done"
# the same with the default-contract leg (its factorisations have other shapes: kept out of the run the syrk averages come from)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ref -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-forms > $OUT/stats_ref.log 2>&1
echo "stats_ref done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-forms --no-reference-rng > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-forms --no-reference-rng > $OUT/pmc_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-forms --no-reference-rng > $OUT/pmc_mfma.log 2>&1
echo "mfma done"
# the default contract (R-stream replay): kernel stats at the metric size with the structured pass of the predictor (default) and
# with the dense pass (GPIRT_RS_LR=2), PMC passes of the pass's products kernel on 128 items
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/replay_stats -- python3 $R/tools/rstream_step.py 8192 1024 > $OUT/replay_stats.log 2>&1
echo "replay stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/replay_pmc_fetch -- python3 $R/tools/rstream_step.py 8192 128 > $OUT/replay_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/replay_pmc_write -- python3 $R/tools/rstream_step.py 8192 128 > $OUT/replay_pmc_write.log 2>&1
echo "replay pmc done"
export GPIRT_RS_LR=2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/replay_dense_stats -- python3 $R/tools/rstream_step.py 8192 1024 > $OUT/replay_dense_stats.log 2>&1
echo "replay dense stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/replay_dense_pmc_fetch -- python3 $R/tools/rstream_step.py 8192 128 > $OUT/replay_dense_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/replay_dense_pmc_write -- python3 $R/tools/rstream_step.py 8192 128 > $OUT/replay_dense_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/replay_dense_pmc_mfma -- python3 $R/tools/rstream_step.py 8192 128 > $OUT/replay_dense_pmc_mfma.log 2>&1
unset GPIRT_RS_LR
echo "replay dense pmc done"
echo "${GPIRT_COMMIT:-unknown}" > $OUT/commit.txt
ls -R $OUT | head -60
