#!/bin/bash
# round 5, GPU call 42: the streaming medians with 32-bit plane offsets (and rows without mirror / clamp in interior strips) against
# 64-bit addresses; operator tests on the product build
set -e
mkdir -p gpurun_out/r05
TOOL=tools/time_ops.py bash tools/ab_time.sh 4096 > gpurun_out/r05/median_offsets_ab.txt 2>&1
grep "==\|median" gpurun_out/r05/median_offsets_ab.txt
python -m pytest tests/test_gpu_operators.py tests/test_gpu_reference.py -x -q 2>&1 | tail -n 2
