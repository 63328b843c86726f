#!/usr/bin/env python3
"""Rebuild the `round5` block (and the top-level keys bench.py reads) of profiles/pmc_traffic.json from the PMC
summaries tools/pmc_summary.py wrote.  Usage: tools/pmc_traffic_update.py <commit> <dir with pmc_summary*.txt>
bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 -- the gfx950 correction of MI355X_MICROARCH.md (HBM / rocprofv3)."""
import json
import os
import re
import sys


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s{2}(\S.*?)\s+n=(\d+)\s+dur=\s*([\d.]+)us\s+(.*)", line)
        if not m:
            continue
        d = out.setdefault(m.group(1).strip(), {"dur_us": []})
        d["dur_us"].append(float(m.group(3)))
        for kv in m.group(4).split():
            k, _, v = kv.partition("=")
            d[k] = float(v)
    return out


def nbytes(k):
    return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0


def block(k):
    b = {"raw_KB": {"FETCH_SIZE": k["FETCH_SIZE"], "WRITE_SIZE": k["WRITE_SIZE"]}, "bytes_per_launch": nbytes(k),
         "l2_hit_share": k["TCC_HIT_sum"] / k["TCC_REQ_sum"],
         "lds_bank_conflict_share": (k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"]) if k.get("SQ_LDS_IDX_ACTIVE") else 0.0,
         "mean_launch_us_under_pmc": sum(k["dur_us"]) / len(k["dur_us"])}
    if k.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        b["mfma_busy_share"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * k["SQ_WAVE_CYCLES"])
    return b


def main(commit, root):
    f64, i8, n512, c64 = (parse(os.path.join(root, n)) for n in
                          ("pmc_summary.txt", "pmc_summary_i8x65.txt", "pmc_summary_n512.txt", "pmc_summary_c64.txt"))
    r5 = {"library_commit": commit,
          "source": "profiles/r05_pmc_summary.txt, r05_pmc_summary_i8x65.txt, r05_pmc_summary_n512.txt, r05_pmc_summary_c64.txt "
                    "(tools/gpu/r5_evidence_b.sh: tools/pmc_pass.sh on the default bench, --products i8x65, --N 512, --dtype c64; one "
                    "rocprofv3 --pmc pass per counter group beside --kernel-trace only; means per dispatch, tagged no-op launches "
                    "included); this block is written by tools/pmc_traffic_update.py from those files",
          "note_mfma_busy": "SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES): per wave cycle; kernels with several wavefronts per SIMD read low",
          "N1024_complex128": {"first_product k_zgemm<64x64>": block(f64["k_zgemm<plain,64x64>"]),
                               "second_product k_zgemm_tri": block(f64["k_zgemm_tri"]),
                               "laplacian_inverse k_solve<9, folded>": block(f64["k_solve"])},
          "N1024_config3_i8x65": {"first_product k_oz_gemm<6,plain>": block(i8["k_oz_gemm<6,plain>"]),
                                  "second_product k_oz_gemm<5/6,fused>": block(i8["k_oz_gemm<5/6,fused>"]),
                                  "slicing k_oz_slice (mean of the two launches)": block(i8["k_oz_slice"])},
          "N512_complex128": {"first_product k_zgemm<32x32>": block(n512["k_zgemm<plain,32x32>"]),
                              "second_product k_zgemm_tri32": block(n512["k_zgemm_tri32"]),
                              "laplacian_inverse k_solve<8>": block(n512["k_solve"])},
          "N1024_complex64": {"first_product k_cgemm_ks": block(c64["k_cgemm_ks"]),
                              "second_product k_cgemm_tri32": block(c64["k_cgemm_tri"]),
                              "laplacian_inverse k_solve<float>": block(c64["k_solve"]),
                              "algorithmic_bytes_per_launch": {"first_product": 3 * 8 * 1024 * 1024, "k_solve": 20 * 1024 * 1024}}}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
    d = json.load(open(path))
    d["round5"] = r5
    d["stamp"] = {"library_commit": commit, "collected": "round 5", "top_level_keys_from": "round5"}
    d["zgemm_plain_bytes_per_launch_N1024"] = nbytes(f64["k_zgemm<plain,64x64>"])
    d["zgemm_tri_bytes_per_launch_N1024"] = nbytes(f64["k_zgemm_tri"])
    d["k_solve_bytes_per_launch_N1024"] = nbytes(f64["k_solve"])
    d["zgemm_plain_bytes_per_launch_N512"] = nbytes(n512["k_zgemm<plain,32x32>"])
    d["zgemm_tri_bytes_per_launch_N512"] = nbytes(n512["k_zgemm_tri32"])
    d["k_solve_bytes_per_launch_N512"] = nbytes(n512["k_solve"])
    d["oz_gemm_i8x65_plain_bytes_per_launch_N1024"] = nbytes(i8["k_oz_gemm<6,plain>"])
    d["oz_gemm_i8x65_fused_bytes_per_launch_N1024"] = nbytes(i8["k_oz_gemm<5/6,fused>"])
    d["complex64_N1024"] = {"cgemm_ks_bytes_per_launch": nbytes(c64["k_cgemm_ks"]), "cgemm_tri_bytes_per_launch": nbytes(c64["k_cgemm_tri"]),
                            "k_solve_float_bytes_per_launch": nbytes(c64["k_solve"]), "from": "round5"}
    json.dump(d, open(path, "w"), indent=1)
    for k in ("N1024_complex128", "N1024_config3_i8x65", "N512_complex128", "N1024_complex64"):
        for name, b in r5[k].items():
            if "bytes_per_launch" in b:
                print("%-20s %-48s %7.1f MB  L2 hit %.2f  LDS conflicts %.2f  MFMA busy %s  %.1f us" % (
                    k, name, b["bytes_per_launch"] / 1e6, b["l2_hit_share"], b["lds_bank_conflict_share"],
                    ("%.2f" % b["mfma_busy_share"]) if "mfma_busy_share" in b else "-", b["mean_launch_us_under_pmc"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
