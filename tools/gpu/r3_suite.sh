#!/bin/bash
# round 3: the whole GPU suite, then the c64 and N=512 bench lines
export TMPDIR=/tmp
out=gpurun_out/r3d; mkdir -p $out
timeout -k 10 200 python -m pytest tests/test_hip_single.py -x -q > $out/pytest_single.txt 2>&1; tail -30 $out/pytest_single.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_single.py > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -60 $out/pytest_gpu.txt; }
tail -3 $out/pytest_gpu.txt
for n in 512 1024 2048; do timeout -k 10 300 python bench.py --dtype c64 --N $n --steps $([ $n = 2048 ] && echo 60 || echo 200) --warmup 10 --cpu-seconds 5 > $out/bench_c64_$n.json 2> $out/bench_c64_$n.err || tail -5 $out/bench_c64_$n.err; python -c "
import json,sys
d=json.loads(open('$out/bench_c64_$n.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c64 N=$n', d['value'], 'its', d['config']['iterations_per_step'], 'gemm1 us', r['avg_launch_us'], 'frac', r['frac'], 'gemm2', r.get('second_product',{}).get('avg_launch_us'), 'solve', r.get('laplacian_inverse',{}).get('avg_launch_us'), 'cpu', d['cpu_baseline'])
"; done
timeout -k 10 300 python bench.py --N 512 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512 K=400', d['value'])"
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1024 K=200', d['value'], d['roofline']['frac'])"
