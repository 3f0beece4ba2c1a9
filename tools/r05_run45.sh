#!/bin/bash
# round 5, GPU call 45: the driver's command (--steps 20 --warmup 5) by steps per lock-step group
set -e
mkdir -p gpurun_out/r05
run() { python3 bench.py --gpus 1 --steps $3 --warmup 5 --workload $1 --step-group $2 --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 steps $3 step-group $2 pairs/s %.1f ms/step %.4f (%.4f-%.4f)' % (d['pairs_per_s'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max']))" || echo "$1 step-group $2 failed"; }
{
for k in 20 40 100; do for g in 1 2 4 8; do run cfg3_4096_gradient $g $k; done; done
} > gpurun_out/r05/step_group_by_steps.txt 2>&1
grep step-group gpurun_out/r05/step_group_by_steps.txt
