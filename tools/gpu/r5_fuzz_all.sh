#!/bin/bash
# the four randomised differential runs against the oracle, small sizes, fresh seeds; every case is a line under gpurun_out/
out=gpurun_out/r05_fuzz; mkdir -p $out
timeout -k 10 300 python - > $out/all_stepper_seed31.jsonl 2>/dev/null <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_stepper_vs_oracle as a
a.main(cases=600, seed=31, sizes=[2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129])
PY
tail -1 $out/all_stepper_seed31.jsonl
timeout -k 10 300 python - > $out/all_backends_seed32.jsonl 2>/dev/null <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_backends_vs_oracle as b
b.main(cases=800, seed=32, sizes=[2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129])
PY
tail -1 $out/all_backends_seed32.jsonl
timeout -k 10 300 python - > $out/all_chains_seed33.jsonl 2>/dev/null <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_trajectory_vs_oracle as c
c.main(cases=60, seed=33, sizes=[48, 64, 96, 100, 128, 160, 192, 256])
PY
tail -1 $out/all_chains_seed33.jsonl
timeout -k 10 300 python - > $out/all_config3_seed34.jsonl 2>/dev/null <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_config3_vs_oracle as d
d.main(cases=60, seed=34, sizes=[64, 128, 192, 256, 320, 384, 448, 512])
PY
tail -1 $out/all_config3_seed34.jsonl
grep -h "\"ok\": false" $out/all_*seed3?.jsonl | cut -c1-300 | head -10
