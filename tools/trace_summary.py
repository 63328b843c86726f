#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 kernel_trace.csv with the stepper's tagged no-op
launches (a launch whose (step, iteration) tag is not due returns at once: ~3-5 us) listed
separately, so that the average duration of EXECUTED launches can be compared with the
HIP-event figure bench.py reports in `roofline.avg_launch_us`."""
import csv
import glob
import sys
from collections import defaultdict


def main(root, noop_us=12.0):
    import os
    f = max(glob.glob(root + "/*/*kernel_trace.csv"), key=os.path.getmtime)
    d = defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in d.values())
    print("%-74s %6s %6s %10s %10s %7s" % ("kernel", "calls", "no-op", "avg_us", "exec_avg", "share"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        guarded = any(t in k for t in ("k_zgemm", "k_solve", "k_update", "k_oz_gemm", "k_oz_slice", "k_cgemm", "k_norm_decide"))
        ex = [x for x in v if not (guarded and x < (6.0 if ("k_oz_slice" in k or "k_norm_decide" in k or "k_solve" in k) else noop_us))]
        print("%-74s %6d %6d %10.1f %10.1f %6.2f%%" % (k[:74], len(v), len(v) - len(ex), sum(v) / len(v),
                                                       sum(ex) / max(len(ex), 1), 100 * sum(v) / total))


if __name__ == "__main__":
    main(sys.argv[1])
