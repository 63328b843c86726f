#!/bin/bash
# round-4 final evidence at HEAD: part A again (suite, smoke, bench lines, kernel traces) + config-3 variants + small-size traces
bash tools/gpu/r4_evidence_a.sh || exit 1
out=gpurun_out/r04_ev
python tools/gpu/r4_config3_variants.py 1024 40 > $out/config3_variants_n1024.jsonl 2> $out/config3_variants_n1024.err; cat $out/config3_variants_n1024.jsonl | cut -c1-260
cd /tmp && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_n512 -- python3 bench.py --N 512 --steps 400 --warmup 20 --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs > $out/bench_n512_under_rocprof.json 2> $out/bench_n512_under_rocprof.err
python3 tools/trace_summary.py $out/prof_n512 > $out/bench_kernel_trace_summary_n512.txt 2>&1; head -5 $out/bench_kernel_trace_summary_n512.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_n2048 -- python3 bench.py --N 2048 --steps 60 --warmup 6 --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs > $out/bench_n2048_under_rocprof.json 2> $out/bench_n2048_under_rocprof.err
python3 tools/trace_summary.py $out/prof_n2048 > $out/bench_kernel_trace_summary_n2048.txt 2>&1; head -5 $out/bench_kernel_trace_summary_n2048.txt
rm -rf $out/prof_n512 $out/prof_n2048
