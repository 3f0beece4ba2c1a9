#!/bin/bash
# round 5, GPU call 52: a lock-step group's finest level as ONE launch (strips as long as the group allows) against instance by instance
set -e
mkdir -p gpurun_out/r05
for g in 8 4 2; do
  echo "== steps per group $g"
  for so in ab/split.so ab/nosplit.so ab/split.so ab/nosplit.so; do
    FLOW2D_HIP_LIB=$PWD/$so python3 bench.py --gpus 1 --steps 40 --warmup 5 --step-group $g --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$so group $g pairs/s %.1f ms/step %.4f check %s' % (d['pairs_per_s'], d['ms_per_step'], d['output_check']['ok']))"
  done
done > gpurun_out/r05/finest_level_split_ab.txt 2>&1
cat gpurun_out/r05/finest_level_split_ab.txt
