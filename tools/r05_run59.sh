#!/bin/bash
# round 5, GPU call 59: more fuzzing of the final library (AUTO, groups and SOR included; against the reference's kernels)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 700 python tools/fuzz_parity.py 5000 901 0 0.3 > gpurun_out/r05/fuzz_auto_long2.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_long2.txt
timeout -k 10 400 python tools/fuzz_reference.py 1500 902 > gpurun_out/r05/fuzz_reference_long2.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_long2.txt
