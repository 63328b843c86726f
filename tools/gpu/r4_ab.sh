#!/bin/bash
# A/B on one box: tools/gpu/r4_ab.sh <outdir> <N> [extra bench args]  -- base library vs the tree's, alternating, 3 rounds
out=$1; N=$2; shift; shift; mkdir -p $out
for r in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export QUFLOW_HIP_LIB=$PWD/tools/ab/libquflow_hip_base.so; else unset QUFLOW_HIP_LIB; fi
    timeout -k 10 120 python bench.py --N $N --no-side-runs --no-config3 --cpu-seconds 0 "$@" > $out/ab_${v}_${N}_$r.json 2> $out/ab_${v}_${N}_$r.err
  done
done
unset QUFLOW_HIP_LIB
python - $out $N <<'PY'
import json,glob,sys
out,N=sys.argv[1],sys.argv[2]
for v in ("base","new"):
    vals=[]
    for f in sorted(glob.glob("%s/ab_%s_%s_*.json"%(out,v,N))):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
            vals.append((d["value"], r.get("avg_launch_us"), (r.get("second_product") or {}).get("avg_launch_us"), (r.get("laplacian_inverse") or {}).get("avg_launch_us")))
        except Exception as e: vals.append(("ERR",str(e)))
    print(v, N, " | ".join(("%.1f (g1 %s)" % (x[0], x[1])) if x[0]!="ERR" else str(x) for x in vals))
PY
