#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py (per-kernel calls / average ns / share).
# Usage (GPU box): tools/kstats.sh <outdir> [bench args...]
export TMPDIR=/tmp
out=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs "$@" > "$out.json" 2> "$out.err"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.05:
        print("%-86s calls %6s avg %9.1f ns  pct %s" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
