#!/bin/bash
# round 5, GPU call 51: long fuzz campaigns on the final tree (strips forced; against the reference's kernels), run_batch8 with one rank
set -e
mkdir -p gpurun_out/r05
timeout -k 10 540 python tools/fuzz_parity.py 3500 801 2 0.35 > gpurun_out/r05/fuzz_strips_long.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_strips_long.txt
timeout -k 10 420 python tools/fuzz_reference.py 1500 802 > gpurun_out/r05/fuzz_reference_long.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_long.txt
bash tools/run_batch8.sh --world 1 > gpurun_out/r05/run_batch8_world1.txt 2>&1; tail -n 2 gpurun_out/r05/run_batch8_world1.txt
