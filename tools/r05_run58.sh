#!/bin/bash
# round 5, GPU call 57: do the waves' memory bursts collide?  starting them in four phases (level solve)
# rows in flight, against the global stores behind the lane mask and against the committed kernel; three rounds on one box
set -e
mkdir -p gpurun_out/r05
for i in 1 2; do bash tools/ab_time.sh 4096 4096 2 5; done > gpurun_out/r05/stagger_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/stagger_ab.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7, "ms"}' | sort | awk '{k=$1" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k]}' | sort
