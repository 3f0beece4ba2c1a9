#!/bin/bash
# round 5, GPU call 37: strip kernel with the exact start-up (no row test in the steady state), tile rule without border terms in
# interior strips: level solve A/B against call 36's builds, then the fused and kernel tests on the product build
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/scalar_trim2_ab.txt 2>&1
cat gpurun_out/r05/scalar_trim2_ab.txt
python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernels.py tests/test_gpu_reference.py -x -q 2>&1 | tail -n 3
