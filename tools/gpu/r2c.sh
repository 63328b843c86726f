#!/bin/bash
out=gpurun_out/r2c; mkdir -p $out
export TMPDIR=/tmp
for v in spread0 spread1 NOSTORE NOGLOAD NOBARRIER; do echo "== $v"; timeout -k 10 120 tools/gemm_time_$v 1024 | tee $out/gemm_time_$v.txt; done
echo "== N=2048 / 512"; timeout -k 10 120 tools/gemm_time_spread0 2048 50 | head -5; timeout -k 10 120 tools/gemm_time_spread1 2048 50 | head -5
timeout -k 10 120 tools/gemm_time_spread0 512 | head -3; timeout -k 10 120 tools/gemm_time_spread1 512 | head -3
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "zgemm or fixedpoint or n64_golden or spot or i8" > $out/pytest_gemm.txt 2>&1 || { echo "pytest failed"; tail -30 $out/pytest_gemm.txt; exit 1; }
tail -2 $out/pytest_gemm.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_default_k20.json 2> $out/bench_default_k20.err || { echo "bench failed"; tail -20 $out/bench_default_k20.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2c/bench_default_k20.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print(json.dumps(d["roofline"], indent=1)[:3500])
print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("value", "casimir_drift", "casimir_drift_f64_run", "spectrum_drift", "spectrum_drift_f64_run")}) for k, v in d.items() if k.startswith("config3") or k == "other_sizes" or k == "cpu_baseline"})
PY
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-config3 --no-side-runs --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20 lean', d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches_timed_with_events'])"
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-config3 --no-side-runs --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=200 lean', d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches_timed_with_events'])"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-config3 --no-side-runs --cpu-seconds 0 > $out/bench_rccl1.json 2> $out/bench_rccl1.err || { echo "rccl bench failed"; tail -20 $out/bench_rccl1.err; }
python -c "import json; d=json.loads(open('gpurun_out/r2c/bench_rccl1.json').read().strip().splitlines()[-1]); print('rccl x1', d['value'], d['config']['gather'])"
