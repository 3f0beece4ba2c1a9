#!/bin/bash
# developer experiment: tile shapes of the LDS kernel (FLOW2D_TILE_VARIANT: 1 = 8x8, 2 = 16x16, 4 = 32x32) against
# the strips (algorithm 2), level solve 10 x 5.   usage (GPU box): bash tools/tile_variants.sh [sizes...]
cd ${GRAFT_REPO_ROOT:-.}
for size in ${@:-192 256 320 384 448 512 640 800}; do
  echo "== ${size}^2 strips"; timeout -k 10 100 python tools/time_sweep.py $size $size 2 2>&1 | grep "level solve"
  for v in 1 2 4; do echo "== ${size}^2 tiles variant $v"; FLOW2D_TILE_VARIANT=$v timeout -k 10 100 python tools/time_sweep.py $size $size 4 2>&1 | grep "level solve"; done
done
