#!/bin/bash
# round 5, GPU call 43: lanes for the launch-bound configuration (rub1 / rub2) and the batch configuration
set -e
mkdir -p gpurun_out/r05
for lanes in 4 6 8 12; do
  for wl in cfg1_rub cfg4_1080p_batch; do
    python3 bench.py --workload $wl --max-lanes $lanes --pipeline $lanes --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl lanes $lanes (used %s) pairs/s %.1f ms/step %.3f' % (d['config'].get('streams_per_gpu'), d['pairs_per_s'], d['ms_per_step']))"
  done
done > gpurun_out/r05/lanes_sweep_cfg1.txt 2>&1
cat gpurun_out/r05/lanes_sweep_cfg1.txt
