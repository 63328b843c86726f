#!/bin/bash
# longer runs than the test suite's: conservation and the stream-K hand-off over many launches
export TMPDIR=/tmp
out=gpurun_out/soak; mkdir -p $out
timeout -k 10 300 python tools/longrun.py 1024 50000 5000 > $out/longrun_n1024.txt 2>&1; tail -1 $out/longrun_n1024.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=1024 50k steps:', round(d['timesteps_per_s'],1), 'steps/s; casimir drift', d['casimir_drift_k234'], 'skew', d.get('skew_hermitian_defect'))"
timeout -k 10 300 python tools/ensemble_rate.py 512 4 20000 | tee $out/ens512_long.jsonl
timeout -k 10 300 python tools/ensemble_rate.py 1024 2 8000 | tee $out/ens1024_long.jsonl
