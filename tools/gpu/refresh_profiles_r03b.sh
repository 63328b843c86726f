#!/bin/bash
# Round 3 evidence, part B: kernel traces, PMC passes, probes, ensembles, long runs, config-3 variants.
out=gpurun_out/refresh3; mkdir -p $out
export TMPDIR=/tmp
tools/kstats.sh $out/kstats > $out/kstats_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats > $out/trace_summary.txt 2>&1; head -8 $out/trace_summary.txt
tools/kstats.sh $out/kstats512 --N 512 --steps 400 --warmup 20 > $out/kstats512_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats512 > $out/trace_summary_n512.txt 2>&1; head -6 $out/trace_summary_n512.txt
python3 tools/iter_timeline.py $out/kstats512 4000 > $out/iter_timeline_n512.txt; python3 tools/iter_timeline.py $out/kstats 2000 > $out/iter_timeline_n1024.txt; head -5 $out/iter_timeline_n512.txt
tools/kstats.sh $out/kstats_c64 --dtype c64 > $out/kstats_c64_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats_c64 > $out/trace_summary_c64.txt 2>&1; head -7 $out/trace_summary_c64.txt
tools/pmc_pass.sh $out/pmc > $out/pmc_pass.log 2>&1; python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt 2>&1; grep -E "^==|k_zgemm|k_solve" $out/pmc_summary.txt
tools/pmc_pass.sh $out/pmc512 --N 512 --steps 200 --warmup 10 > $out/pmc512_pass.log 2>&1; python3 tools/pmc_summary.py $out/pmc512 > $out/pmc_summary_n512.txt 2>&1; grep -E "^==|k_zgemm|k_solve" $out/pmc_summary_n512.txt
for n in 512 1024 2048; do echo "== N=$n" >> $out/solve_probe.txt; timeout -k 10 60 tools/solve_probe $n >> $out/solve_probe.txt 2>&1; done
timeout -k 10 120 tools/gemm_time 1024 > $out/gemm_time_n1024.txt 2>&1; timeout -k 10 120 tools/gemm_time 512 400 > $out/gemm_time_n512.txt 2>&1; timeout -k 10 120 tools/gemm_time 256 400 > $out/gemm_time_n256.txt 2>&1
for n in 512 1024; do timeout -k 10 300 python tools/ensemble_rate.py $n 1,2,4 300 >> $out/ensemble_rates.jsonl; done; cat $out/ensemble_rates.jsonl
timeout -k 10 300 python tools/longrun.py 2048 10000 1000 > $out/longrun_n2048.txt 2>&1; tail -1 $out/longrun_n2048.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['timesteps_per_s'], d['roofline'])"
timeout -k 10 600 python tools/gpu/r3_hybrid.py 1024 40 > $out/config3_variants_n1024.jsonl 2> $out/config3_variants.err; cat $out/config3_variants_n1024.jsonl | cut -c1-260
