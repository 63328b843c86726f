#!/bin/bash
# Everything under profiles/r02_*: the GPU suite, smoke, the bench lines, kernel trace, PMC passes, probes, ensemble rates, long run.
# Usage: gpurun --timeout 1200 -- bash tools/gpu/refresh_profiles.sh ; then copy from gpurun_out/refresh/ (profiles/README_r02.md)
out=gpurun_out/refresh; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest_gpu.txt; exit 1; }
tail -2 $out/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
grep -h "enstrophy drift per step" $out/pytest_gpu.txt
timeout -k 10 120 python -m pytest tests/test_hip_parity.py -q -s -k "enstrophy_drift" 2>&1 | grep "drift per step"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20.json 2> $out/bench_driver_form_k20.err || { echo "bench failed"; tail -20 $out/bench_driver_form_k20.err; exit 1; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20_b.json 2>> $out/bench_driver_form_k20.err
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { echo "bench failed"; exit 1; }
for n in 512 2048; do timeout -k 10 300 python bench.py --N $n --steps $([ $n = 512 ] && echo 400 || echo 60) --warmup 10 --no-config3 --cpu-seconds 8 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err; done
timeout -k 10 300 python bench.py --ic B --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
timeout -k 10 300 python bench.py --fixed-iters 4 --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
timeout -k 10 300 python bench.py --compsum --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
timeout -k 10 300 python bench.py --products i8x6 --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
timeout -k 10 300 python bench.py --products i8 --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
python - <<'PY'
import json
for l in open("gpurun_out/refresh/bench_lines.jsonl"):
    d = json.loads(l); c = d["config"]; r = d.get("roofline") or {}
    print("N=%d ic=%s products=%s fixed=%s compsum=%s: %.1f steps/s  its %.3f  gemm1 %.1f us exec_frac %.3f" % (c["N"], c["ic"], c["products"], c["fixed_iters"], c["compsum"], d["value"], c["iterations_per_step"], r.get("avg_launch_us", 0), r.get("executed_frac", 0)))
for f in ("bench_driver_form_k20", "bench_default"):
    d = json.loads(open("gpurun_out/refresh/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["executed_frac"], d["roofline"].get("whole_step", {}).get("frac"))
PY
for n in 512 1024; do timeout -k 10 300 python tools/ensemble_rate.py $n 1,2,4 300 >> $out/ensemble_rates.jsonl; done; cat $out/ensemble_rates.jsonl
tools/kstats.sh $out/kstats > $out/kstats_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats > $out/trace_summary.txt 2>&1; cat $out/trace_summary.txt | head -12
tools/pmc_pass.sh $out/pmc > $out/pmc_pass.log 2>&1; python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt 2>&1; grep -E "^==|k_zgemm|k_solve" $out/pmc_summary.txt
timeout -k 10 60 tools/solve_probe 1024 > $out/solve_probe_n1024.txt 2>&1
timeout -k 10 120 tools/gemm_time 1024 > $out/gemm_time_n1024.txt 2>&1; cat $out/gemm_time_n1024.txt
timeout -k 10 300 python tools/longrun.py 2048 10000 1000 > $out/longrun_n2048.txt 2>&1; tail -1 $out/longrun_n2048.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['timesteps_per_s'], d['roofline'])"

o2=$out/k20trace; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o2 -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --no-side-runs > $o2.json 2> $o2.err; python3 tools/trace_timeline.py $o2 "k_zgemm<" 10 > $out/timeline_k20.txt; tail -8 $out/timeline_k20.txt
