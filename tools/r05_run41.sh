#!/bin/bash
# round 5, GPU call 41: how far apart two non-plain instructions may be to share one raised-priority run (issue_priority.py GAP)
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/priority_gap_ab.txt 2>&1
cat gpurun_out/r05/priority_gap_ab.txt
