#!/bin/bash
# round 5, GPU call 70: two / three / six input rows in flight as register sets named by the step's ring position (global stores, one
# block per step: the compiler's waits become vmcnt(7) / (13) / (31)) against the committed kernel; four rounds on one box
set -e
mkdir -p gpurun_out/r05
for i in 1 2; do bash tools/ab_time.sh 4096 4096 2 5; done > gpurun_out/r05/row_sets_global_store_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/row_sets_global_store_ab.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7}' | sort | awk '{k=$1" "$2" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k], "ms"}' | sort
