#!/bin/bash
out=gpurun_out/r05_b
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -q -k "i8 or config3 or config5 or int8 or full_size or fault or hybrid" > $out/pytest_i8.txt 2>&1; echo "i8 subset rc=$?"; tail -5 $out/pytest_i8.txt
timeout -k 10 300 python tools/gpu/r5_icb.py 1024 200 f64,i8x65,i8x6 > $out/icb_n1024.jsonl 2> $out/icb.err; tail -2 $out/icb.err
QUFLOW_HIP_GEMM=auto timeout -k 10 800 python -m pytest tests -m gpu -q > $out/pytest_gpu_under_auto_products.txt 2>&1; echo "auto rc=$?"; tail -6 $out/pytest_gpu_under_auto_products.txt
