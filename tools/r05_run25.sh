#!/bin/bash
# round 5, GPU call 25: s_setprio around the strip kernel's unpairable instructions (csrc/issue_priority.py), with and without
# packed arithmetic: level solve A/B, then correctness of the candidates
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/issue_priority_ab.txt 2>&1
cat gpurun_out/r05/issue_priority_ab.txt
