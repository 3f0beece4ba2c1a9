#!/bin/bash
# round 5, GPU call 27: the issue-priority filter (and no packed arithmetic) on EVERY kernel file, whole-pipeline rates + operators
set -e
mkdir -p gpurun_out/r05
WLS="cfg3_4096_gradient cfg2_1024_grey cfg4_1080p_batch cfg1_rub" bash tools/ab_bench.sh ab/base.so ab/allprio.so ab/allprio_nopk.so > gpurun_out/r05/all_priority_bench_ab.txt 2>&1
cat gpurun_out/r05/all_priority_bench_ab.txt
TOOL=tools/time_ops.py bash tools/ab_time.sh 4096 4096 > gpurun_out/r05/all_priority_ops_ab.txt 2>&1 || true
tail -n 60 gpurun_out/r05/all_priority_ops_ab.txt
