#!/bin/bash
# round 5, GPU call 28: lanes in flight against the pipelined rate with the new strip kernel
set -e
mkdir -p gpurun_out/r05
for lanes in 2 3 4 6 8; do
  for wl in cfg3_4096_gradient cfg2_1024_grey; do
    python3 bench.py --workload $wl --pipeline $lanes --max-lanes $lanes --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl lanes $lanes pairs/s %.1f ms/step %.3f' % (d['pairs_per_s'], d['ms_per_step']))"
  done
done > gpurun_out/r05/lanes_sweep.txt 2>&1
cat gpurun_out/r05/lanes_sweep.txt
