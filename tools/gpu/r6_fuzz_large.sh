#!/bin/bash
# round 6: the randomised differential harnesses (tests/fuzz_*.py) drawn at LARGE sizes -- both sides of every kernel-selection
# rule of DESIGN 3.0 (512 | 513, 767 | 768, 895 | 896, 959 | 960, 1024 | 1025, 2175 | 2176) with the stepper's options and hooks,
# the Laplacian backends and the other steppers; with every device allocation fenced (QUFLOW_HIP_DEBUG_GUARD=1).
# Usage: gpurun --timeout 1200 -- bash tools/gpu/r6_fuzz_large.sh
export TMPDIR=/tmp
export QUFLOW_HIP_DEBUG_GUARD=1
out=gpurun_out/r06_fuzz_large; mkdir -p $out
timeout -k 10 420 python - > $out/stepper_large_seed71.jsonl 2>$out/stepper_large_seed71.err <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_stepper_vs_oracle as a
a.main(cases=160, seed=71 + int(__import__("os").environ.get("FUZZ_SEED_OFFSET", "0")), sizes=[384, 500, 511, 512, 513, 640, 767, 768, 800, 895, 896, 959, 960, 1000, 1024, 1025, 1088])
import quflow_amd; quflow_amd.release_contexts(); print('guard zones: %d fenced, %d damaged %s' % quflow_amd.guard_report())
PY
tail -2 $out/stepper_large_seed71.jsonl
timeout -k 10 420 python - > $out/backends_large_seed72.jsonl 2>$out/backends_large_seed72.err <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_backends_vs_oracle as b
b.main(cases=160, seed=72 + int(__import__("os").environ.get("FUZZ_SEED_OFFSET", "0")), sizes=[300, 511, 512, 513, 767, 768, 769, 1000, 1024, 1025, 1151, 1152])
import quflow_amd; quflow_amd.release_contexts(); print('guard zones: %d fenced, %d damaged %s' % quflow_amd.guard_report())
PY
tail -2 $out/backends_large_seed72.jsonl
timeout -k 10 300 python - > $out/config3_large_seed73.jsonl 2>$out/config3_large_seed73.err <<'PY'
import sys; sys.path.insert(0, "tests")
import fuzz_config3_vs_oracle as d
d.main(cases=40, seed=73 + int(__import__("os").environ.get("FUZZ_SEED_OFFSET", "0")), sizes=[768, 832, 896, 960, 1024, 1088, 1280, 1536])
import quflow_amd; quflow_amd.release_contexts(); print('guard zones: %d fenced, %d damaged %s' % quflow_amd.guard_report())
PY
tail -2 $out/config3_large_seed73.jsonl
grep -h "guard" $out/*.err | head -5
grep -h "\"ok\": false" $out/*_seed7?.jsonl | cut -c1-300 | head -10
exit 0
