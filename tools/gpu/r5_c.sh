#!/bin/bash
out=gpurun_out/r05_c
mkdir -p $out
timeout -k 10 800 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; echo "default rc=$?"; tail -8 $out/pytest_gpu.txt
timeout -k 10 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; tail -3 $out/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c/bench_default.json"))
r = d["roofline"]
print("value", d["value"], "no-prewarm", d["config"]["value_without_prewarm"])
print("kernel:", r["kernel"])
print("second:", r["second_product"]["kernel"], r["second_product"]["tile_share"], r["second_product"]["avg_launch_us"])
print("solve:", r["laplacian_inverse"]["kernel"], r["laplacian_inverse"]["avg_launch_us"])
print("ranks:", d["config"]["ranks"])
print("config3:", d["config3_lowprecision_products"]["value"], d["config3_lowprecision_products"]["vs_fp64_headline"], d["config3_lowprecision_products"].get("N2048", {}).get("vs_fp64_same_size"))
print("other:", {k: (v["value"], v["whole_step_frac"], v["kernels"]) for k, v in d["other_sizes"].items()})
PY
