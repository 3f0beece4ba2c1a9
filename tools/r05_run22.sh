#!/bin/bash
# round 5, GPU call 22: share and clustering of special instructions against the SIMD's issue rate (tools/ubench/gen_issue_mix.py)
set -e
mkdir -p gpurun_out/r05
timeout -k 5 240 ./build_ubench/issue_mix 3000 > gpurun_out/r05/issue_mix.txt
tail -n 5 gpurun_out/r05/issue_mix.txt
