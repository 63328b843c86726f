#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py, one pass per counter group (gfx950 slot limits:
# SQ 8, TCC 4 with FETCH_SIZE=3 / WRITE_SIZE=2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage (on the GPU box): tools/pmc_pass.sh <outdir> [bench args...]
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE GRBM_GUI_ACTIVE"
)
i=0
for c in "${groups[@]}"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pass$i" -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs "$@" > "$out/pass$i.json" 2> "$out/pass$i.err" || { echo "pass $i failed"; tail -5 "$out/pass$i.err"; }
done
find "$out" -name "*counter_collection.csv" | head
