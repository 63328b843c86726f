#!/bin/bash
# round 6: the four randomised differential runs against the oracle with fresh seeds -- the sizes now span the 4-entry chunks of k_solve (N <= 256) and their edge (255, 256, 257)
out=gpurun_out/r06_fuzz; mkdir -p $out
timeout -k 10 280 python - > $out/all_stepper_seed61.jsonl 2>/dev/null <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_stepper_vs_oracle as a
a.main(cases=400, seed=61 + int(__import__("os").environ.get("FUZZ_SEED_OFFSET", "0")), sizes=[2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129, 192, 255, 256, 257])
PY
tail -1 $out/all_stepper_seed61.jsonl
timeout -k 10 280 python - > $out/all_backends_seed62.jsonl 2>/dev/null <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_backends_vs_oracle as b
b.main(cases=600, seed=62 + int(__import__("os").environ.get("FUZZ_SEED_OFFSET", "0")), sizes=[2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129, 192, 255, 256, 257])
PY
tail -1 $out/all_backends_seed62.jsonl
timeout -k 10 280 python - > $out/all_chains_seed63.jsonl 2>/dev/null <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_trajectory_vs_oracle as c
c.main(cases=60, seed=63 + int(__import__("os").environ.get("FUZZ_SEED_OFFSET", "0")), sizes=[48, 64, 96, 100, 128, 160, 192, 256])
PY
tail -1 $out/all_chains_seed63.jsonl
timeout -k 10 280 python - > $out/all_config3_seed64.jsonl 2>/dev/null <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_config3_vs_oracle as d
d.main(cases=60, seed=64 + int(__import__("os").environ.get("FUZZ_SEED_OFFSET", "0")), sizes=[64, 128, 192, 256, 320, 384, 448, 512])
PY
tail -1 $out/all_config3_seed64.jsonl
grep -h "\"ok\": false" $out/all_*seed6?.jsonl | cut -c1-300 | head -10
