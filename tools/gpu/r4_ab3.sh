#!/bin/bash
# A/B/C on one box: tools/gpu/r4_ab3.sh <outdir> <N> <libs...> -- [bench args]
out=$1; N=$2; shift; shift; libs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do libs+=($1); shift; done; shift
mkdir -p $out
for r in 1 2 3; do
  for v in "${libs[@]}"; do
    QUFLOW_HIP_LIB=$PWD/tools/ab/libquflow_hip_$v.so timeout -k 10 120 python bench.py --N $N --no-side-runs --no-config3 --cpu-seconds 0 "$@" > $out/ab_${v}_${N}_$r.json 2> $out/ab_${v}_${N}_$r.err
  done
done
python - $out $N "${libs[@]}" <<'PY'
import json,glob,sys
out,N=sys.argv[1],sys.argv[2]
for v in sys.argv[3:]:
    vals=[]
    for f in sorted(glob.glob("%s/ab_%s_%s_*.json"%(out,v,N))):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
            vals.append("%.1f (g1 %.2f)"%(d["value"], r.get("avg_launch_us") or 0))
        except Exception as e: vals.append("ERR "+str(e))
    print("%-10s N=%s  %s" % (v, N, " | ".join(vals)))
PY
