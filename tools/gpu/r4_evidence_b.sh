#!/bin/bash
# round-4 evidence, part B: PMC passes (separate --pmc runs, kernel trace only beside them), fp64 default + tile order 8 (TCC / traffic only)
export TMPDIR=/tmp
out=gpurun_out/r04_pmc; mkdir -p $out
cd /tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_pass.sh $out/f64 > $out/f64_passes.txt 2>&1
python3 tools/pmc_summary.py $out/f64 > $out/pmc_summary.txt 2>&1; head -60 $out/pmc_summary.txt
# tile order 8 (blocked): L2 hit rate / traffic of k_zgemm_tri
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1)); mkdir -p $out/order8
  QUFLOW_HIP_TRI_ORDER=8 timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/order8/pass$i" -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs > "$out/order8/pass$i.json" 2> "$out/order8/pass$i.err" || echo "order8 pass $i failed"
done
python3 tools/pmc_summary.py $out/order8 > $out/pmc_summary_order8.txt 2>&1; grep -A12 "k_zgemm_tri" $out/pmc_summary_order8.txt | head -40
# keep the summaries, drop the raw per-dispatch csv (tens of MB)
find $out -name "*counter_collection.csv" -size +1M -delete; find $out -name "*kernel_trace.csv" -size +1M -delete
du -sh $out
