#!/bin/bash
# round-5 evidence, part B: PMC passes (separate --pmc runs, kernel trace only beside them): fp64 default, config 3's
# products, N = 512; then config 5's long run in fp64 beside the i8x65 one already in profiles/
export TMPDIR=/tmp
out=gpurun_out/r05_pmc; mkdir -p $out
cd /tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_pass.sh $out/f64 > $out/f64_passes.txt 2>&1
python3 tools/pmc_summary.py $out/f64 > $out/pmc_summary.txt 2>&1; head -40 $out/pmc_summary.txt
bash tools/pmc_pass.sh $out/i8x65 --products i8x65 > $out/i8x65_passes.txt 2>&1
python3 tools/pmc_summary.py $out/i8x65 > $out/pmc_summary_i8x65.txt 2>&1
bash tools/pmc_pass.sh $out/n512 --N 512 --steps 400 --warmup 20 > $out/n512_passes.txt 2>&1
python3 tools/pmc_summary.py $out/n512 > $out/pmc_summary_n512.txt 2>&1
bash tools/pmc_pass.sh $out/c64 --dtype c64 > $out/c64_passes.txt 2>&1
python3 tools/pmc_summary.py $out/c64 > $out/pmc_summary_c64.txt 2>&1
find $out -name "*counter_collection.csv" -size +1M -delete; find $out -name "*kernel_trace.csv" -size +1M -delete
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
du -sh $out
mkdir -p gpurun_out/r05_ev
timeout -k 10 500 python tools/longrun.py 2048 10000 1000 > gpurun_out/r05_ev/longrun_n2048_10k_steps.json 2> gpurun_out/r05_ev/longrun_n2048_10k_steps.err
tail -c 1500 gpurun_out/r05_ev/longrun_n2048_10k_steps.json
