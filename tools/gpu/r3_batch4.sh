#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3e; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_hip_single.py tests/test_native_comm.py -x -q -m gpu > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest.txt; }
tail -3 $out/pytest.txt
for n in 512 1024 2048; do timeout -k 10 300 python bench.py --dtype c64 --N $n --steps $([ $n = 2048 ] && echo 60 || echo 200) --warmup 10 --cpu-seconds 0 > $out/bench_c64_$n.json 2> $out/bench_c64_$n.err || tail -5 $out/bench_c64_$n.err; python -c "
import json,sys
d=json.loads(open('$out/bench_c64_$n.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c64 N=$n', d['value'], 'gemm1 us', r['avg_launch_us'], 'frac', r['frac'], 'gemm2', r.get('second_product',{}).get('avg_launch_us'))
"; done
timeout -k 10 600 python tools/gpu/r3_hybrid.py 1024 40 > $out/hybrid_1024.jsonl 2> $out/hybrid_1024.err; cat $out/hybrid_1024.jsonl; tail -3 $out/hybrid_1024.err
timeout -k 10 300 python tools/gpu/r3_hybrid.py 256 10 > $out/hybrid_256.jsonl 2> $out/hybrid_256.err; cat $out/hybrid_256.jsonl
