#!/bin/bash
export TMPDIR=/tmp
echo "default"; timeout -k 10 200 python tools/gpu/r2u_one.py ""
echo "GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python tools/gpu/r2u_one.py ""
echo "GPU_MAX_HW_QUEUES=16"; GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python tools/gpu/r2u_one.py ""
echo "GPU_MAX_HW_QUEUES=8, full bench K=20"; GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python tools/gpu/r2t_one.py plain
