#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3m; mkdir -p $out
timeout -k 10 300 python tools/longrun.py 1024 50000 5000 > $out/longrun_n1024_50k.txt 2>&1; tail -1 $out/longrun_n1024_50k.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c128 N=1024', d['timesteps_per_s'], d['skew_hermitian_defect'], d['casimir_drift_k234'], d['chunks'][-1])"
timeout -k 10 300 python tools/longrun.py 1024 50000 5000 c64 > $out/longrun_n1024_50k_c64.txt 2>&1; tail -1 $out/longrun_n1024_50k_c64.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c64 N=1024', d['timesteps_per_s'], d['skew_hermitian_defect'], d['casimir_drift_k234'], d['chunks'][-1])"
timeout -k 10 300 python tools/longrun.py 768 30000 5000 c64 > $out/longrun_n768_30k_c64.txt 2>&1; tail -1 $out/longrun_n768_30k_c64.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c64 N=768', d['timesteps_per_s'], d['skew_hermitian_defect'], d['casimir_drift_k234'], d['chunks'][-1])"
timeout -k 10 300 python tools/longrun.py 2048 5000 1000 c64 > $out/longrun_n2048_5k_c64.txt 2>&1; tail -1 $out/longrun_n2048_5k_c64.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c64 N=2048', d['timesteps_per_s'], d['skew_hermitian_defect'], d['casimir_drift_k234'], d['chunks'][-1], d['roofline']['whole_step_frac'])"
