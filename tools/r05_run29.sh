#!/bin/bash
# round 5, GPU call 29: three input rows in flight against two (strip kernel), level solve
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/rows_in_flight_ab.txt 2>&1
cat gpurun_out/r05/rows_in_flight_ab.txt
