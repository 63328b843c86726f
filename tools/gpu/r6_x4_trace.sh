#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06_align; mkdir -p $out
cd /tmp && cd $GRAFT_REPO_ROOT
for mode in 0 1; do
  export QUFLOW_HIP_DEBUG_GUARD=$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/prof_x4_g$mode -- python3 tools/ensemble_rate.py 512 4 300 > $out/x4_g$mode.json 2> $out/x4_g$mode.err
  cat $out/x4_g$mode.json
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/prof_x4_g$mode/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
q = collections.Counter((r.get("Queue_Id"), r.get("Stream_Id")) for r in rows)
print("kernels per (Queue_Id, Stream_Id):", dict(q))
byq = collections.defaultdict(collections.Counter)
for r in rows: byq[r.get("Queue_Id")][r["Kernel_Name"][:40]] += 1
for k, v in byq.items(): print(" queue", k, dict(v.most_common(4)))
PY
done
rm -rf $out/prof_x4_g*
