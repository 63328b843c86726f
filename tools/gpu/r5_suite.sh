#!/bin/bash
# the whole GPU suite with the default products and under QUFLOW_HIP_GEMM=auto (config 3's products from N = 1024)
out=gpurun_out/r05_suite
mkdir -p $out
timeout -k 10 800 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; echo "default rc=$?"; tail -4 $out/pytest_gpu.txt
QUFLOW_HIP_GEMM=auto timeout -k 10 800 python -m pytest tests -m gpu -q > $out/pytest_gpu_under_auto_products.txt 2>&1; echo "auto rc=$?"; tail -6 $out/pytest_gpu_under_auto_products.txt
