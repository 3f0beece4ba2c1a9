#!/bin/bash
# round 5, GPU call 62: which kernels the small configurations spend their time in now (groups of 32, four lanes, graph replay)
set -e
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
for wl in cfg1_rub cfg2_1024_grey; do
  rm -rf gpurun_out/trace_g32_$wl
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_g32_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg --steps 256 --repeats 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r05/trace_g32_$wl.err)
  python3 tools/summarize_trace.py gpurun_out/trace_g32_$wl/*/*kernel_trace.csv 14 > gpurun_out/r05/${wl}_groups_of_32_by_grid.txt
  rm -rf gpurun_out/trace_g32_$wl
  cat gpurun_out/r05/${wl}_groups_of_32_by_grid.txt
done
