#!/usr/bin/env python3
"""Where an iteration's time goes, from a rocprofv3 kernel_trace.csv: for every kernel name the mean duration of
its EXECUTED launches and the mean gap between the previous kernel's end and its start (dependent launches on one
stream: the gap is the kernel boundary), over the last `tail` launches of the run (steady state).
Usage: tools/iter_timeline.py <rocprof outdir> [tail launches]"""
import csv
import glob
import os
import re
import sys


def main(root, tail=3000):
    f = max(glob.glob(root + "/*/*kernel_trace.csv"), key=os.path.getmtime)
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[-tail:]
    stats = {}
    prev_end = None
    for s, e, name in rows:
        m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", name)
        short = m.group(1)[:70] if m else name[:70]
        d = (e - s) / 1e3
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        prev_end = e
        executed = d > 4.0            # tagged launches that are not due return at once
        st = stats.setdefault((short, executed), [0, 0.0, 0.0])
        st[0] += 1
        st[1] += d
        st[2] += gap
    span = (rows[-1][1] - rows[0][0]) / 1e3
    print("last %d launches span %.1f us" % (len(rows), span))
    for (short, executed), (n, d, g) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %s n=%5d  avg %7.2f us  gap before %6.2f us  share %5.1f%%" %
              (short, "run " if executed else "noop", n, d / n, g / n, 100.0 * (d + g) / span))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3000)
