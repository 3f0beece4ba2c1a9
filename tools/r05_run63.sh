#!/bin/bash
# round 5, GPU call 63: four strip bodies (border selects of the touched borders only) against two, by the planner's cost ratio
set -e
mkdir -p gpurun_out/r05
for size in "4096 4096" "1920 1080" "1024 1024"; do
  echo "#### $size"
  for i in 1 2; do bash tools/ab_time.sh $size 2 5; done | grep "==\|constancy" | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7}' | sort | awk '{k=$1" "$2" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k], "ms"}' | sort
done > gpurun_out/r05/edge_xy_split_ab.txt 2>&1
cat gpurun_out/r05/edge_xy_split_ab.txt
