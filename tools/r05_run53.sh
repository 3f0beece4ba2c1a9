#!/bin/bash
# round 5, GPU call 53: SQ counters of the strip kernel after the issue-priority build (where the wave cycles go)
set -e
mkdir -p gpurun_out/r05
bash tools/pmc_sq.sh 4096 > gpurun_out/r05/fused_sq_counters.txt 2>&1
cat gpurun_out/r05/fused_sq_counters.txt
