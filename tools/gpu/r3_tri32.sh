#!/bin/bash
# round 3: k_zgemm_tri32 -- parity subset, kernel timings, bench at N=512
export TMPDIR=/tmp
out=gpurun_out/r3b; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -x -q -k "tri32 or variants_agree or triangle_protocols or n64_golden or spot or large or bit_identical or fault or chunking or fixedpoint_products" > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -60 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
timeout -k 10 100 tools/gemm_time 512 400 > $out/gemm_time_512.txt 2>&1; cat $out/gemm_time_512.txt
timeout -k 10 100 tools/gemm_time 256 400 > $out/gemm_time_256.txt 2>&1; cat $out/gemm_time_256.txt
for i in 1 2; do timeout -k 10 200 python bench.py --N 512 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512 K=400', d['value'], d['roofline']['avg_launch_us'], d['config']['value_without_prewarm'])"; done
QUFLOW_HIP_TRI32=0 timeout -k 10 200 python bench.py --N 512 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512 K=400 TRI32=0', d['value'], d['roofline']['avg_launch_us'])"
QUFLOW_HIP_TRI32_SPLIT=2,2 timeout -k 10 200 python bench.py --N 512 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512 K=400 split 2,2', d['value'], d['roofline']['avg_launch_us'])"
timeout -k 10 200 python bench.py --N 256 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=256 K=400', d['value'])"
QUFLOW_HIP_TRI32=0 timeout -k 10 200 python bench.py --N 256 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=256 K=400 TRI32=0', d['value'])"
