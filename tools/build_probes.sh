#!/bin/bash
# (re)build the diagnostic harnesses that include the library's kernel sources (run from the repo root; no GPU needed)
F="-O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1"
H=/opt/rocm/bin/hipcc
$H $F tools/gemm_time.hip -o tools/gemm_time || exit 1
$H $F -DQF_STAMP_LIGHT tools/tri_probe.hip -o tools/tri_probe_light || exit 1
$H -O3 -std=c++17 --offload-arch=gfx950 tools/solve_probe.hip -o tools/solve_probe || exit 1
$H -O2 --offload-arch=gfx950 tools/bf16_split_probe.hip -o tools/bf16_split_probe || exit 1
ls -la tools/gemm_time tools/tri_probe_light tools/solve_probe tools/bf16_split_probe
