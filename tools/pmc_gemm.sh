#!/bin/bash
# PMC passes over one plain NT product (tools/gemm_one.py): where the GEMM main loop's wave cycles go.
# Run on the GPU box from the repo root:  bash tools/pmc_gemm.sh [M N K]
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/gemm_one.py "$@" > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_gemm/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f64_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:28s} {tot[k]/n[k]:16.4g}  (avg per launch, {n[k]} launches)")
PY
