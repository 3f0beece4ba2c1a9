#!/bin/bash
# round 5, GPU call 44: steps per lock-step group
set -e
mkdir -p gpurun_out/r05
run() { python3 bench.py --workload $1 --step-group $2 --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 step-group $2 pairs/s %.1f ms/step %.3f single %.2f ms  used GiB %s' % (d['pairs_per_s'], d['ms_per_step'], 1000/d['pairs_per_s_single'], d['device_memory']['used_gib']))" || echo "$1 step-group $2 failed"; }
{
for g in 4 8; do run cfg3_4096_gradient $g; done
for g in 2 4 8; do run cfg3_4096_grey $g; done
for g in 1 2; do run cfg5_8192_grey $g; done
for g in 2 4; do run cfg3_4096_sor $g; done
} > gpurun_out/r05/step_group_sweep3.txt 2>&1
grep step-group gpurun_out/r05/step_group_sweep3.txt
