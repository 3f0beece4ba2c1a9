#!/bin/bash
# round 5, GPU call 50: GPU suite on the final tree
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_final3.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_final3.txt; exit 1; }
tail -n 2 gpurun_out/r05/gpu_tests_final3.txt
