#!/bin/bash
# final check of the round: the GPU suite, smoke, the driver-form bench line and the complex64 lines
export TMPDIR=/tmp
out=gpurun_out/r3final; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest_gpu.txt; exit 1; }
tail -2 $out/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20.json 2> $out/bench_driver_form_k20.err || { echo "bench failed"; tail -20 $out/bench_driver_form_k20.err; exit 1; }
python -c "import json; d=json.loads(open('$out/bench_driver_form_k20.json').read().strip().splitlines()[-1]); print('K=20', d['value'], d['roofline']['frac'], d['config'].get('value_without_prewarm'))"
rm -f $out/bench_c64.jsonl
for n in 512 768 1024 2048; do timeout -k 10 300 python bench.py --dtype c64 --N $n --steps $([ $n = 2048 ] && echo 60 || echo 200) --warmup 10 --cpu-seconds 6 >> $out/bench_c64.jsonl 2>> $out/bench_c64.err; done
python - <<'PY'
import json
for l in open("gpurun_out/r3final/bench_c64.jsonl"):
    d = json.loads(l); c = d["config"]; r = d.get("roofline") or {}
    print("c64 N=%d: %.1f steps/s  its %.3f  gemm1 %.1f us frac %.3f  gemm2 %s solve %s cpu %s" % (
        c["N"], d["value"], c["iterations_per_step"], r.get("avg_launch_us", 0), r.get("frac", 0), (r.get("second_product") or {}).get("avg_launch_us"), (r.get("laplacian_inverse") or {}).get("avg_launch_us"), d["cpu_baseline"]["value"]))
PY
tools/kstats.sh $out/kstats_c64 --dtype c64 > $out/kstats_c64_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats_c64 > $out/trace_summary_c64.txt 2>&1; head -7 $out/trace_summary_c64.txt
