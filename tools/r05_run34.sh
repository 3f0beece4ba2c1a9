#!/bin/bash
# round 5, GPU call 34: the planner's border / interior cost ratio with the new strip kernel
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/edge_cost_ab.txt 2>&1
cat gpurun_out/r05/edge_cost_ab.txt
