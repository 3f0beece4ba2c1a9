#!/bin/bash
# round 5, GPU call 68: one more fuzz campaign on the final library (new seeds)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 800 python tools/fuzz_parity.py 6000 1001 0 0.3 > gpurun_out/r05/fuzz_auto_long3.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_long3.txt
timeout -k 10 300 python tools/fuzz_parity.py 1500 1002 2 0.35 > gpurun_out/r05/fuzz_strips_long3.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_strips_long3.txt
