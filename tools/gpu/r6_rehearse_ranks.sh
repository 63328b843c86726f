#!/bin/bash
# round 6: REHEARSAL of bench.py's N-rank flow on the build's one-GPU box (QUFLOW_BENCH_REHEARSAL=share-gpu: the ranks share the
# GPU and gather over gloo -- RCCL refuses two ranks on one device).  What runs for real: the launcher (self-launch AND
# torch.distributed.run, the driver's command), one process per rank with its own device context, cpu pinning inside the box's
# cpuset, the collectives' warm-up, the timed region with its barriers, the gather of diagnostics, max-over-ranks timing, the
# JSON line.  What it is NOT: an N-GPU measurement (the line says so in config.rehearsal; 4 trajectories share one GPU here).
# Usage: gpurun --timeout 900 -- bash tools/gpu/r6_rehearse_ranks.sh
export TMPDIR=/tmp
out=gpurun_out/r06_rehearsal; mkdir -p $out
export QUFLOW_BENCH_REHEARSAL=share-gpu
for n in 2 4; do
  timeout -k 10 300 python bench.py --gpus $n --steps 20 --warmup 5 > $out/selflaunch_$n.json 2> $out/selflaunch_$n.err; rc=$?
  echo "self-launch $n ranks: rc $rc"; grep "bench.py: rank" $out/selflaunch_$n.err | cut -c1-200
  [ $rc = 0 ] || { tail -20 $out/selflaunch_$n.err; exit $rc; }
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 20 --warmup 5 > $out/torchrun_4.json 2> $out/torchrun_4.err; rc=$?
echo "torch.distributed.run 4 ranks: rc $rc"; grep "bench.py: rank" $out/torchrun_4.err | cut -c1-200
[ $rc = 0 ] || { tail -20 $out/torchrun_4.err; exit $rc; }
python - <<'PY'
import json
for f in ("selflaunch_2", "selflaunch_4", "torchrun_4"):
    d = json.loads(open("gpurun_out/r06_rehearsal/%s.json" % f).read().strip().splitlines()[-1]); c = d["config"]
    print(f, "n_gpus", d["n_gpus"], "value %.1f" % d["value"], "per rank", [round(x) for x in c["per_rank_timesteps_per_s"]], "gather", c["gather"]["backend"],
          "rows ok", c["gather"]["gathered_rows_ok"], "seeds", c["gather"]["seeds_gathered"], "distinct devices", c["distinct_devices_bound"], "| ", c["rehearsal"])
PY
