#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3c; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_abi_and_host.py -x -q -m gpu -k "n64_golden or chunking or spot or diagnostics or contract or options or ensemble or tol or yielding or golden" > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
o2=$out/k20trace; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o2 -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --no-side-runs > $o2.json 2> $o2.err
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3c/k20trace/*/*kernel_trace.csv')[0]
d={}
for r in csv.DictReader(open(f)):
    for k in ('k_inner2','k_call_begin','k_mirror_lower'):
        if k in r['Kernel_Name']: d.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print(k, len(v), 'avg %.1f min %.1f'%(sum(v)/len(v), min(v)))
PY
for i in 1 2; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20', d['value'], d['roofline']['avg_launch_us'])"; done
