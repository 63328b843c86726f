#!/usr/bin/env python3
"""Duration of every EXECUTED launch of one kernel over time, from a rocprofv3 kernel_trace.csv:
shows how the kernel's duration moves through warm-up, idle gaps and the timed region (GPU clock state).
Usage: tools/trace_timeline.py <rocprof outdir> [kernel substring] [bucket_ms]"""
import csv
import glob
import os
import sys


def main(root, key="k_zgemm<", bucket_ms=10.0):
    f = max(glob.glob(root + "/*/*kernel_trace.csv"), key=os.path.getmtime)
    rows = []
    allk = []
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        allk.append((s, e))
        if key in r["Kernel_Name"] and "tri" not in r["Kernel_Name"] and (e - s) > 12000:
            rows.append((s, (e - s) / 1e3))
    rows.sort()
    allk.sort()
    t0 = allk[0][0]
    # idle gaps > 50 us between consecutive kernels (any kernel)
    gaps = []
    last_end = allk[0][1]
    for s, e in allk[1:]:
        if s - last_end > 50000:
            gaps.append(((last_end - t0) / 1e6, (s - last_end) / 1e3))
        last_end = max(last_end, e)
    print("idle gaps > 50 us (at ms, length us):", ["%.1f:%.0f" % g for g in gaps][-30:])
    buckets = {}
    for s, d in rows:
        b = int((s - t0) / 1e6 / bucket_ms)
        buckets.setdefault(b, []).append(d)
    for b in sorted(buckets):
        v = buckets[b]
        print("t=%7.1f ms  n=%3d  avg %6.1f us  min %6.1f  max %6.1f" % (b * bucket_ms, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], a[1] if len(a) > 1 else "k_zgemm<", float(a[2]) if len(a) > 2 else 10.0)
