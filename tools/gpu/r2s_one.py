import os, sys, time, json, io, contextlib
sys.path.insert(0, ".")
import bench
which = sys.argv[1]
noop = lambda *a, **k: {"value": 0, "roofline_executed_frac_first_product": 0}
patch = {"other": {"other_size_run": noop}, "config3": {"config3_side_run": lambda *a, **k: {}},
         "passes": {"fixed_iteration_run": lambda *a, **k: {}, "instrumented_pass": lambda *a, **k: ({"gemm1": {"avg_s": 1e-4}, "gemm2": {"avg_s": 1e-4}, "poisson": {"avg_s": 1e-5}}, {})},
         "none": {}}[which]
for k, v in patch.items():
    setattr(bench, k, v)
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0"] + sys.argv[2:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
r = d["replicas_per_gpu"]["N512_x4"]
print("first main() without '%s' %s:" % (which, sys.argv[7:]), "value %.0f" % d["value"], "x4 %.0f ratio %.3f" % (r["sum_timesteps_per_s"], r["ratio"]), flush=True)
