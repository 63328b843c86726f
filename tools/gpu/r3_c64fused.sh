#!/bin/bash
# round 3: complex64 with the fused step end -- its tests, then the bench lines with and without it
export TMPDIR=/tmp
out=gpurun_out/r3i; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_single.py tests/test_hip_faults.py "tests/test_hip_parity.py::test_fixedpoint_products_tri32" -x -q > $out/pytest_single.txt 2>&1; tail -30 $out/pytest_single.txt
for fused in 1 0; do for n in 768 1024 2048; do
QUFLOW_HIP_GEMM2=$([ $fused = 1 ] && echo tri || echo full) timeout -k 10 300 python bench.py --dtype c64 --N $n --steps $([ $n = 2048 ] && echo 60 || echo 200) --warmup 10 --cpu-seconds 0 > $out/bench_c64_${n}_f$fused.json 2> $out/bench_c64_${n}_f$fused.err || tail -5 $out/bench_c64_${n}_f$fused.err; python -c "
import json,sys
d=json.loads(open('$out/bench_c64_${n}_f$fused.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c64 fused=$fused N=$n', d['value'], 'its', d['config']['iterations_per_step'], 'gemm1 us', r['avg_launch_us'], 'frac', r['frac'], 'gemm2', r.get('second_product',{}).get('avg_launch_us'), 'solve', r.get('laplacian_inverse',{}).get('avg_launch_us'))
"; done; done
