#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in i8x65 f64; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$p -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs --products $p > $out/bench_prof_$p.json 2> $out/bench_prof_$p.err
  python3 tools/trace_summary.py $out/prof_$p > $out/trace_summary_$p.txt 2>&1
  cat $out/trace_summary_$p.txt | head -12
  rm -rf $out/prof_$p/*/*_agent_info.csv
done
du -sh $out
