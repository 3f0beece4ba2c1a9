#!/bin/bash
# round 5, GPU call 35: a long fuzz campaign on the final library
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1000 python tools/fuzz_parity.py 6000 601 0 0.3 > gpurun_out/r05/fuzz_auto_long.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_long.txt
