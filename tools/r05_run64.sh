#!/bin/bash
# round 5, GPU call 64: the driver's 20 steps over four lanes: groups of 8 + 8 + 4, of 5 x 4, of 4 x 5
set -e
mkdir -p gpurun_out/r05
run() { python3 bench.py --gpus 1 --steps $3 --warmup 5 --workload $1 --step-group $2 --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 steps $3 step-group $2 pairs/s %.1f ms/step %.4f (%.4f-%.4f)' % (d['pairs_per_s'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max']))" || echo "$1 step-group $2 failed"; }
{
for rep in 1 2; do for g in 8 5 4 10; do run cfg3_4096_gradient $g 20; done; done
} > gpurun_out/r05/step_group_20_steps.txt 2>&1
grep step-group gpurun_out/r05/step_group_20_steps.txt
