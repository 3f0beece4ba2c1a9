#!/bin/bash
# round 5, GPU call 54: what the strip waves wait for (s_waitcnt is 16 % of their cycles now): non-temporal stores / loads, three rows in flight
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/memory_wait_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/memory_wait_ab.txt | awk '/==/{n=$2} /constancy/{print n, $2, $7}'
