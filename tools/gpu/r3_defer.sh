#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3h; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -x -q -k "deferred or triangle_protocols or n64_golden or spot or large or bit_identical or fault or chunking or variants_agree or general_branch or solve_driver or advance_with_diagnostics or ensemble" > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -60 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
for n in 512 256 128; do for d in 1 0; do QUFLOW_HIP_DEFER=$d timeout -k 10 200 python bench.py --N $n --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n DEFER=$d', d['value'], d['config']['iterations_per_step'])"; done; done
QUFLOW_HIP_DEFER=1 timeout -k 10 200 python bench.py --N 512 --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512 K=20 DEFER=1', d['value'])"
QUFLOW_HIP_DEFER=0 timeout -k 10 200 python bench.py --N 512 --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512 K=20 DEFER=0', d['value'])"
timeout -k 10 200 python tools/ensemble_rate.py 512 1,2,4 300
