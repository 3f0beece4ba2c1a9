#!/bin/bash
# round 5, GPU call 31: which build of the strip kernel for which launch -- the rule (lone up to one workgroup per CU), always
# paired, always lone: whole-pipeline rates and the lone pair's latency
set -e
mkdir -p gpurun_out/r05
for rep in 1 2; do
for kind in auto paired lone; do
  if [ $kind = auto ]; then unset FLOW2D_FUSED_KIND; else export FLOW2D_FUSED_KIND=$kind; fi
  echo "== strip kernel build: $kind"
  WLS="cfg3_4096_gradient cfg2_1024_grey cfg4_1080p_batch cfg1_rub" bash tools/ab_bench.sh ab/dual.so | awk 'NR<=4'
done
done > gpurun_out/r05/strip_kernel_kind_ab.txt 2>&1
cat gpurun_out/r05/strip_kernel_kind_ab.txt
