#!/bin/bash
# round 5, GPU call 47: the batch leg before the main job against behind it (driver's command), and the main value without it
set -e
mkdir -p gpurun_out/r05
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1: main pairs/s %.1f  batch %s  h2d %.1f  launch_ms %s' % (d['pairs_per_s'], (d.get('batch') or {}).get('pairs_per_s'), d['pairs_per_s_incl_h2d'], d['roofline']['avg_launch_ms']))"; }
{
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline 2>/dev/null | show "batch leg first"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline --batch-leg-last 2>/dev/null | show "batch leg last "
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | show "no batch leg   "
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline 2>/dev/null | show "batch leg first"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline --batch-leg-last 2>/dev/null | show "batch leg last "
python3 bench.py --workload cfg4_1080p_batch --steps 64 --no-pmc --no-cpu-baseline --no-reference-baseline --no-host-entry-leg 2>/dev/null | show "cfg4 on its own"
} > gpurun_out/r05/batch_leg_first_ab.txt 2>&1
cat gpurun_out/r05/batch_leg_first_ab.txt
