#!/bin/bash
# round 2, call B: full GPU parity suite on the reworked tri epilogue / call protocol, A/B bench, tri phase probe
out=gpurun_out/r2b; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest_gpu.txt; exit 1; }
tail -3 $out/pytest_gpu.txt
for rep in 1 2; do
  for lib in new r01; do
    if [ $lib = r01 ]; then export QUFLOW_HIP_LIB=$PWD/tools/ab/libquflow_hip_r01.so; else unset QUFLOW_HIP_LIB; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-config3 --cpu-seconds 0 > $out/bench_${lib}_k20_$rep.json 2> $out/bench_${lib}_k20_$rep.err || { echo "bench failed"; tail -5 $out/bench_${lib}_k20_$rep.err; exit 1; }
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-config3 --cpu-seconds 0 > $out/bench_${lib}_k200_$rep.json 2> $out/bench_${lib}_k200_$rep.err || { echo "bench failed"; exit 1; }
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-config3 --cpu-seconds 0 --no-kernel-events > $out/bench_${lib}_k20ne_$rep.json 2> $out/bench_${lib}_k20ne_$rep.err || { echo "bench failed"; exit 1; }
  done
done
unset QUFLOW_HIP_LIB
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2b/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print("%-44s %8.1f steps/s  gemm1 %6.1f us  gemm2 %6.1f us  its %.3f" % (f.split("/")[-1], d["value"], r.get("avg_launch_us", 0), (r.get("second_product") or {}).get("avg_launch_us", 0), d["config"]["iterations_per_step"]))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 60 tools/tri_probe 1024 > $out/tri_probe_nonfused.txt 2>&1; QF_FUSED=1 timeout -k 10 60 tools/tri_probe 1024 > $out/tri_probe_fused.txt 2>&1
QF_NOSTAMPS=1 QF_FUSED=1 timeout -k 10 60 tools/tri_probe 1024 > $out/tri_probe_fused_nostamps.txt 2>&1
cat $out/tri_probe_fused.txt; grep "tri " $out/tri_probe_fused_nostamps.txt
# K-loop ablations of the first product (round-1 builds of the stamped kernel; timing only)
for v in BASE NOBARRIER NOGLOAD NOSTORE NOSUMS; do echo "== $v"; QF_NOSTAMPS=1 timeout -k 10 60 tools/zgemm_probe_$v 1024 0 2>&1 | grep -E "^rep [2-4]"; done
QF_PHASES=1 timeout -k 10 60 tools/zgemm_probe 1024 0 2>&1 | grep -E "phase|K-tile  |wave total|prologue|epilogue" 
timeout -k 10 120 python bench.py --N 512 --steps 200 --warmup 20 --no-config3 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512', d['value'], d['roofline']['avg_launch_us'], d['roofline']['second_product']['avg_launch_us'])"
timeout -k 10 120 python bench.py --N 2048 --steps 40 --warmup 5 --no-config3 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2048', d['value'], d['roofline']['avg_launch_us'], d['roofline']['second_product']['avg_launch_us'])"
