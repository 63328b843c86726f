#!/bin/bash
out=gpurun_out/r04_soak; mkdir -p $out
python tools/longrun.py 1024 50000 5000 > $out/longrun_n1024_50k_steps.json 2> $out/longrun_n1024_50k_steps.progress
python -c "import json; d=json.load(open('$out/longrun_n1024_50k_steps.json')); print('1024 x 50k', d['timesteps_per_s'], d['casimir_drift_k234'], d['skew_hermitian_defect'], d['roofline']['whole_step_frac'])"
QUFLOW_HIP_GEMM=i8x65 python tools/longrun.py 2048 10000 1000 > $out/longrun_n2048_10k_steps_i8x65.json 2> $out/longrun_n2048_10k_steps_i8x65.progress
python -c "import json; d=json.load(open('$out/longrun_n2048_10k_steps_i8x65.json')); print('2048 i8x65 x 10k', d['timesteps_per_s'], d['casimir_drift_k234'], d['skew_hermitian_defect'], d['trace'])"
python tools/longrun.py 512 100000 10000 > $out/longrun_n512_100k_steps.json 2> $out/longrun_n512_100k_steps.progress
python -c "import json; d=json.load(open('$out/longrun_n512_100k_steps.json')); print('512 x 100k', d['timesteps_per_s'], d['casimir_drift_k234'], d['skew_hermitian_defect'])"
