#!/usr/bin/env python3
"""Rebuild the `round<R>` block (and the top-level keys bench.py reads) of profiles/pmc_traffic.json from the PMC
summaries tools/pmc_summary.py wrote.  Usage: tools/pmc_traffic_update.py <commit> <dir with pmc_summary*.txt> [round, default 5]
(round >= 6 also reads pmc_summary_n2048.txt: BASELINE config 5's size)
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


def main(commit, root, rnd=5):
    f64, i8, n512, c64 = (parse(os.path.join(root, n)) for n in
                          ("pmc_summary.txt", "pmc_summary_i8x65.txt", "pmc_summary_n512.txt", "pmc_summary_c64.txt"))
    tag = "r%02d" % rnd
    r5 = {"library_commit": commit,
          "source": "profiles/%s_pmc_summary.txt, %s_pmc_summary_i8x65.txt, %s_pmc_summary_n512.txt, %s_pmc_summary_c64.txt " % (tag, tag, tag, tag) +
                    "(tools/pmc_pass.sh on the default bench, --products i8x65, --N 512, --dtype c64; one "
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
    blocks = ["N1024_complex128", "N1024_config3_i8x65", "N512_complex128", "N1024_complex64"]
    n2048_path = os.path.join(root, "pmc_summary_n2048.txt")
    if rnd >= 6 and os.path.exists(n2048_path):
        n2048 = parse(n2048_path)
        r5["N2048_complex128"] = {"first_product k_zgemm<64x64>": block(n2048["k_zgemm<plain,64x64>"]),
                                  "second_product k_zgemm_tri": block(n2048["k_zgemm_tri"]),
                                  "laplacian_inverse k_solve<17, folded>": block(n2048["k_solve"]),
                                  "algorithmic_bytes_per_launch": {"first_product": 3 * 16 * 2048 * 2048, "k_solve": 40 * 2048 * 2048}}
        d["zgemm_plain_bytes_per_launch_N2048"] = nbytes(n2048["k_zgemm<plain,64x64>"])
        d["zgemm_tri_bytes_per_launch_N2048"] = nbytes(n2048["k_zgemm_tri"])
        d["k_solve_bytes_per_launch_N2048"] = nbytes(n2048["k_solve"])
        blocks.append("N2048_complex128")
    d["round%d" % rnd] = r5
    d["stamp"] = {"library_commit": commit, "collected": "round %d" % rnd, "top_level_keys_from": "round%d" % rnd}
    d["zgemm_plain_bytes_per_launch_N1024"] = nbytes(f64["k_zgemm<plain,64x64>"])
    d["zgemm_tri_bytes_per_launch_N1024"] = nbytes(f64["k_zgemm_tri"])
    d["k_solve_bytes_per_launch_N1024"] = nbytes(f64["k_solve"])
    d["zgemm_plain_bytes_per_launch_N512"] = nbytes(n512["k_zgemm<plain,32x32>"])
    d["zgemm_tri_bytes_per_launch_N512"] = nbytes(n512["k_zgemm_tri32"])
    d["k_solve_bytes_per_launch_N512"] = nbytes(n512["k_solve"])
    d["oz_gemm_i8x65_plain_bytes_per_launch_N1024"] = nbytes(i8["k_oz_gemm<6,plain>"])
    d["oz_gemm_i8x65_fused_bytes_per_launch_N1024"] = nbytes(i8["k_oz_gemm<5/6,fused>"])
    d["complex64_N1024"] = {"cgemm_ks_bytes_per_launch": nbytes(c64["k_cgemm_ks"]), "cgemm_tri_bytes_per_launch": nbytes(c64["k_cgemm_tri"]),
                            "k_solve_float_bytes_per_launch": nbytes(c64["k_solve"]), "from": "round%d" % rnd}
    json.dump(d, open(path, "w"), indent=1)
    for k in blocks:
        for name, b in r5[k].items():
            if "bytes_per_launch" in b:
                print("%-20s %-48s %7.1f MB  L2 hit %.2f  LDS conflicts %.2f  MFMA busy %s  %.1f us" % (
                    k, name, b["bytes_per_launch"] / 1e6, b["l2_hit_share"], b["lds_bank_conflict_share"],
                    ("%.2f" % b["mfma_busy_share"]) if "mfma_busy_share" in b else "-", b["mean_launch_us_under_pmc"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5)
