#!/usr/bin/env python3
"""What one stepper call carries besides its iterations: the LAST call of a rocprofv3 kernel trace (bench.py's timed region
when nothing runs after it) laid out kernel by kernel -- every launch that is not one of the three iteration kernels, with the
idle time in front of it, and the totals.  Usage: tools/call_overhead.py <rocprof outdir>"""
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


def main(root):
    f = max(glob.glob(root + "/*/*kernel_trace.csv"), key=os.path.getmtime)
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))))
    # the last k_call_begin opens the last call
    starts = [i for i, k in enumerate(ks) if k[2].startswith("k_call_begin")]
    i0 = starts[-1]
    call = ks[i0:]
    main3 = ("k_zgemm<", "k_zgemm_tri", "k_solve<")
    t_first, t_last = call[0][0], call[-1][1]
    busy_main = sum(e - s for s, e, n in call if n.startswith(main3))
    n_main = sum(1 for s, e, n in call if n.startswith(main3))
    other = [(s, e, n) for s, e, n in call if not n.startswith(main3)]
    print("last call: %d launches over %.1f us; the three iteration kernels: %d launches, %.1f us" % (
        len(call), (t_last - t_first) / 1e3, n_main, busy_main / 1e3))
    prev_end = {}
    last_end = None
    tot_other = tot_gap = 0.0
    for idx, (s, e, n) in enumerate(call):
        gap = 0.0 if last_end is None else max(0, s - last_end) / 1e3
        if not n.startswith(main3) or gap > 3.0:
            print("  +%9.1f us  %-48s %7.1f us   idle before it %6.1f us" % ((s - t_first) / 1e3, n, (e - s) / 1e3, gap))
        if not n.startswith(main3):
            tot_other += (e - s) / 1e3
        tot_gap += gap
        last_end = e if last_end is None else max(last_end, e)
    print("other kernels %.1f us, idle gaps %.1f us -> %.1f us per call besides the iterations" % (tot_other, tot_gap, tot_other + tot_gap))


if __name__ == "__main__":
    main(sys.argv[1])
