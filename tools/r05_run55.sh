#!/bin/bash
# round 5, GPU call 55: the strip kernel with every row folded onto cache-resident rows (timing probe): what memory costs it now
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/compute_only_probe.txt 2>&1
grep "==\|constancy" gpurun_out/r05/compute_only_probe.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7, "ms"}'
