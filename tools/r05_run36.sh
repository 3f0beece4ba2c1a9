#!/bin/bash
# round 5, GPU call 36: strip kernel without the zero-increment branch and the interior row clamp: level solve A/B, quick parity
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/scalar_trim_ab.txt 2>&1
cat gpurun_out/r05/scalar_trim_ab.txt
