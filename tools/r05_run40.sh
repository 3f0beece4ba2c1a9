#!/bin/bash
# round 5, GPU call 40: where AUTO hands a level from the LDS tiles to the strips, with the faster strip kernel (pipelined rates)
set -e
mkdir -p gpurun_out/r05
for rep in 1 2; do
for px in 360000 250000 160000 90000 40000; do
  export FLOW2D_TILED_MAX_PIXELS=$px
  echo "== tiles up to $px pixels"
  WLS="cfg3_4096_gradient cfg1_rub cfg2_1024_grey" bash tools/ab_bench.sh ab/devfull.so 2>/dev/null | awk 'NR<=3'
done
done > gpurun_out/r05/tiled_max_pixels_ab.txt 2>&1
cat gpurun_out/r05/tiled_max_pixels_ab.txt
