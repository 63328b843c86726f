#!/bin/bash
# round 6: k replicas on one GPU for k beyond the four pipes (profiles/r06_x4_hardware_queues.txt, section 8)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06_align
for a in "512 1,2,3,4,5,6,8" "256 1,2,4,8" "1024 1,2,3,4"; do
  set -- $a
  timeout -k 10 250 python tools/ensemble_rate.py $1 $2 300 2>/dev/null | python3 -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('N=$1', ' '.join('x%d %.0f (%.2f)' % (r['replicas_on_one_gpu'], r['sum_timesteps_per_s'], r['vs_single']) for r in rows))"
done | tee gpurun_out/r06_align/replica_counts.txt
