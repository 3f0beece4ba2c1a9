#!/bin/bash
# round 5, GPU call 24: the strip kernel's own row-step instruction streams replayed (tools/ubench/gen_replay.py)
set -e
mkdir -p gpurun_out/r05
timeout -k 5 120 ./build_ubench/replay 2000 > gpurun_out/r05/replay.txt
cat gpurun_out/r05/replay.txt
