#!/bin/bash
# round 5, GPU call 71: the strip kernel's loads and stores alone (no arithmetic: timing probe) with two / three / six rows in flight
set -e
mkdir -p gpurun_out/r05
for i in 1 2; do bash tools/ab_time.sh 4096 4096 2 5; done > gpurun_out/r05/memory_only_row_sets.txt 2>&1
grep "==\|constancy" gpurun_out/r05/memory_only_row_sets.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7}' | sort | awk '{k=$1" "$2" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k], "ms"}' | sort
