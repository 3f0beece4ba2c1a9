#!/bin/bash
# round 5, GPU call 30: the round's measurement pass again, with the plain-instruction strip kernel (profiles/r05_*)
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh bench trace trace_default > gpurun_out/r05/measure_final.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_final.txt; exit 1; }
cat gpurun_out/r05/measure_final.txt
