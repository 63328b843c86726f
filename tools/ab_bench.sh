#!/bin/bash
# A/B of the second-product variants on the GPU box: tools/ab_bench.sh <outdir> [N ...]
out=$1; shift
mkdir -p "$out"
for m in full tri; do for n in "$@"; do
  QUFLOW_HIP_GEMM2=$m timeout -k 10 200 python bench.py --N $n --cpu-seconds 0 > "$out/bench_${m}_$n.json" 2> "$out/bench_${m}_$n.err"
done; done
python - "$out" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "steps/s %.1f  ms/step %.4f  gemm avg us %.1f  its %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["config"]["iterations_per_step"]))
    except Exception as e:
        print(f, "ERR", e)
PY
