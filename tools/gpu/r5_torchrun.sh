#!/bin/bash
# the driver's N > 1 form with one rank on the 1-GPU box: torch.distributed.run, nccl (= RCCL) and the torch-free route, beside the plain run
out=gpurun_out/r05_torchrun; mkdir -p $out
show() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); c=d['config']
print('$2', round(d['value'],1), 'without prewarm', c.get('value_without_prewarm'), c['gather'], c.get('timed_region_ms_rank0'))"; }
A="--steps 20 --warmup 5 --cpu-seconds 0 --no-side-runs --no-config3"
for rep in 1 2; do
timeout -k 10 300 python bench.py $A > $out/plain_$rep.json 2> $out/plain_$rep.err; show $out/plain_$rep.json plain
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep bench.py --gpus 1 $A > $out/torch_$rep.json 2> $out/torch_$rep.err; show $out/torch_$rep.json torchrun-nccl
QUFLOW_BENCH_GATHER=native timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$rep bench.py --gpus 1 $A > $out/native_$rep.json 2> $out/native_$rep.err; show $out/native_$rep.json torchrun-native
done
