#!/bin/bash
# round 5, GPU call 23: the strip kernel's row step without packed arithmetic and / or without lane shifts (timing probes)
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/plain_stream_probe.txt 2>&1
cat gpurun_out/r05/plain_stream_probe.txt
