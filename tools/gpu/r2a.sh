#!/bin/bash
# round 2, call A: launch-boundary probe, parity of the swizzled LDS image, A/B against the round-1 library
out=gpurun_out/r2a; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 120 tools/launch_probe > $out/launch_probe.txt 2>&1 || { echo "launch_probe failed"; tail -5 $out/launch_probe.txt; exit 1; }
echo "probe done"
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "zgemm or fixedpoint or n64_golden or spot" > $out/pytest_gemm.txt 2>&1 || { echo "pytest failed"; tail -30 $out/pytest_gemm.txt; exit 1; }
tail -3 $out/pytest_gemm.txt
for rep in 1 2; do
  for lib in new r01; do
    if [ $lib = r01 ]; then export QUFLOW_HIP_LIB=$PWD/tools/ab/libquflow_hip_r01.so; else unset QUFLOW_HIP_LIB; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-config3 --cpu-seconds 0 > $out/bench_${lib}_k20_$rep.json 2> $out/bench_${lib}_k20_$rep.err || { echo "bench failed"; tail -5 $out/bench_${lib}_k20_$rep.err; exit 1; }
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-config3 --cpu-seconds 0 > $out/bench_${lib}_k200_$rep.json 2> $out/bench_${lib}_k200_$rep.err || { echo "bench failed"; exit 1; }
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-config3 --cpu-seconds 0 --no-kernel-events > $out/bench_${lib}_k200ne_$rep.json 2> $out/bench_${lib}_k200ne_$rep.err || { echo "bench failed"; exit 1; }
  done
done
unset QUFLOW_HIP_LIB
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2a/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print("%-44s %8.1f steps/s  gemm1 %6.1f us  gemm2 %6.1f us  its %.3f" % (f.split("/")[-1], d["value"], r.get("avg_launch_us", 0), (r.get("second_product") or {}).get("avg_launch_us", 0), d["config"]["iterations_per_step"]))
    except Exception as e:
        print(f, "ERR", e)
PY
# LDS counters of the new build
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc/pass2 -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 > $out/pmc_pass2.json 2> $out/pmc_pass2.err || { echo "pmc pass failed"; tail -5 $out/pmc_pass2.err; }
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt 2>&1; grep -E "k_zgemm|k_solve" $out/pmc_summary.txt
cat $out/launch_probe.txt
