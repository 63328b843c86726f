import os, sys, time, json, io, contextlib
sys.path.insert(0, ".")
mode = sys.argv[1]
t_start = time.time()
import bench
def run(argv):
    sys.argv = ["bench.py"] + argv
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    r = d.get("replicas_per_gpu", {}).get("N512_x4")
    return d["value"], (r["sum_timesteps_per_s"], r["ratio"]) if r else None
if mode == "sleep":
    time.sleep(15)
elif mode == "tiny_first":
    print("tiny", run(["--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-config3"]))
elif mode == "import_torch":
    import torch
print(mode, "t=%.1f" % (time.time() - t_start), run(["--steps", "20", "--warmup", "5", "--cpu-seconds", "0"]), flush=True)
