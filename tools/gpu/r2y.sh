#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2y
out=gpurun_out/r2y/k20
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --no-side-runs > $out.json 2> $out.err
python3 tools/trace_timeline.py $out "k_zgemm<" 10 > gpurun_out/r2y/timeline_k20.txt; tail -40 gpurun_out/r2y/timeline_k20.txt
for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20', d['value'], d['roofline']['avg_launch_us'])"; done
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=200', d['value'], d['roofline']['avg_launch_us'])"
