#!/bin/bash
# round 5, GPU call 33: a register budget for the strip kernel below what two waves need (room for other lanes' kernels)
set -e
mkdir -p gpurun_out/r05
WLS="cfg3_4096_gradient cfg3_4096_grey cfg4_1080p_batch" bash tools/ab_bench.sh ab/dev.so ab/v224.so ab/v208.so > gpurun_out/r05/vgpr_budget_ab.txt 2>&1
cat gpurun_out/r05/vgpr_budget_ab.txt
