#!/bin/bash
out=gpurun_out/r04m; mkdir -p $out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "tri or fused or n64_golden or vs_oracle_large or headline or fixedpoint_products or faults or ensemble" > $out/pytest_tri.txt 2>&1; tail -3 $out/pytest_tri.txt
./tools/gemm_time 1024 > $out/gemm_time_1024.txt 2>&1; grep "triangle" $out/gemm_time_1024.txt
QUFLOW_HIP_SK_SCHED=0 ./tools/gemm_time 1024 > $out/gemm_time_1024_sched0.txt 2>&1; grep "triangle" $out/gemm_time_1024_sched0.txt
QF_FUSED=1 ./tools/tri_probe_light 1024 > $out/tri_probe_light_1024.txt 2>&1; sed -n 9,24p $out/tri_probe_light_1024.txt
for KH in 0 33 34 35 36 37; do
  for r in 1 2; do
  QUFLOW_HIP_SK_HEAD_KT=$KH timeout -k 10 120 python bench.py --no-side-runs --no-config3 --cpu-seconds 0 > $out/bench_KH${KH}_$r.json 2> $out/bench_KH${KH}_$r.err
  python - $out/bench_KH${KH}_$r.json $KH <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("KH", sys.argv[2], "%.1f"%d["value"])
PY
  done
done
QUFLOW_HIP_SK_SCHED=0 timeout -k 10 120 python bench.py --no-side-runs --no-config3 --cpu-seconds 0 > $out/bench_sched0.json 2> $out/bench_sched0.err; python -c "
import json; d=json.loads(open('$out/bench_sched0.json').read().strip().splitlines()[-1]); print('sched0 %.1f'%d['value'])"
