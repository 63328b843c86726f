#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python - <<'PY'
import os, sys, time, json, io, contextlib
sys.path.insert(0, ".")
import bench
def run(tag, patch):
    saved = {k: getattr(bench, k) for k in patch}
    for k, v in patch.items():
        setattr(bench, k, v)
    sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    for k, v in saved.items():
        setattr(bench, k, v)
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    r = d["replicas_per_gpu"]["N512_x4"]
    print(tag, "value %.0f" % d["value"], "x4 %.0f ratio %.3f" % (r["sum_timesteps_per_s"], r["ratio"]), flush=True)
noop = lambda *a, **k: {"value": 0, "roofline_executed_frac_first_product": 0}
run("all side runs", {})
run("no other sizes", {"other_size_run": noop})
run("no config3", {"config3_side_run": lambda *a, **k: {}})
run("no fixed-iteration, no instrumented pass", {"fixed_iteration_run": lambda *a, **k: {}, "instrumented_pass": lambda *a, **k: ({"gemm1": {"avg_s": 1e-4}, "gemm2": {"avg_s": 1e-4}, "poisson": {"avg_s": 1e-5}}, {})})
PY
