#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/pmc_pass.sh output): per kernel name, mean counter
value per dispatch (+ mean duration from the kernel trace of the same pass)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    if "k_zgemm_tri32" in name:
        return "k_zgemm_tri32"
    if "k_zgemm_tri" in name:
        return "k_zgemm_tri"
    if "k_cgemm_tri" in name:
        return "k_cgemm_tri"
    if "k_cgemm_ks" in name:
        return "k_cgemm_ks"
    if "k_cgemm32" in name or "k_cgemm" in name:
        flat = name.replace(" ", "")
        return ("k_cgemm32" if "k_cgemm32" in name else "k_cgemm") + ("<EPI>" if "<true," in flat else "<plain>")
    if "k_oz_gemm" in name:
        flat = name.replace(" ", "")
        digits = flat.split("k_oz_gemm<")[1].split(",")[0] if "k_oz_gemm<" in flat else "?"
        targs = flat.split("k_oz_gemm<")[1].split(">")[0].split(",") if "k_oz_gemm<" in flat else []
        fused = len(targs) > 1 and targs[1] == "true"
        layout = ("/" + targs[2]) if len(targs) > 2 and targs[2] != targs[0] else ""
        return "k_oz_gemm<%s%s,%s>" % (digits, layout, "fused" if fused else "plain")
    if "k_oz_slice" in name:
        return "k_oz_slice"
    for key in ("k_zgemm", "k_solve", "k_update", "k_max_rows", "k_row_abs_sum", "k_inner", "k_sum_partials",
                "k_build_factors", "k_lap_table", "copyBuffer", "fillBuffer"):
        if key in name:
            if key == "k_zgemm":
                args = name.split("k_zgemm<")[1].split(">")[0].replace(" ", "").split(",")
                return "k_zgemm<%s,%s>" % ("EPI" if args[4] == "true" else "plain", "x".join(args[:2]))
            return key
    return name[:40]


def main(root):
    for cc in sorted(glob.glob(os.path.join(root, "pass*", "*", "*counter_collection.csv"))):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(cc)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = defaultdict(list)
        kt = glob.glob(os.path.join(os.path.dirname(cc), "*kernel_trace.csv"))
        if kt:
            for r in csv.DictReader(open(kt[0])):
                dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("==", os.path.relpath(cc, root))
        for k in sorted(agg):
            n = max(len(v) for v in agg[k].values())
            d = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
            cols = "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items()))
            print("  %-16s n=%-4d dur=%8.1fus  %s" % (k, n, d, cols))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_r01")
