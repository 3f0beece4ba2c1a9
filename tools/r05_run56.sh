#!/bin/bash
# round 5, GPU call 56: hand-written loads / stores / waits in the strip kernel (2 and 3 row sets in flight): timing, then parity
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/manual_wait_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/manual_wait_ab.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7, "ms"}'
for v in manual2 manual3; do
  echo "== tests with ab/$v.so"
  FLOW2D_HIP_LIB=$PWD/ab/$v.so timeout -k 10 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -n 2
done
