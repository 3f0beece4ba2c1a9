#!/bin/bash
# round 5, GPU call 61: kernel traces of the final tree (the default command as the driver runs it; the eager single-stream traces)
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh trace trace_default > gpurun_out/r05/measure_traces_final2.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_traces_final2.txt; exit 1; }
grep "fused_outer_kernel<5, [01], true, false, false>  *131072" gpurun_out/r05/measure_traces_final2.txt | head
