#!/bin/bash
# A/B timing of solver builds on ONE GPU box (boxes differ by a few %, so variants must share a run):
# every ab/*.so is loaded in turn through FLOW2D_HIP_LIB (the in-tree library is never touched) and timed with
# tools/time_sweep.py, twice.  Variants are built with
#   make -C cuda-flow2d_amd/csrc BUILD=build_<name> LIB=$PWD/ab/<name>.so EXTRA=-D...
# usage (on the GPU box): bash tools/ab_time.sh [time_sweep args]      or   TOOL=tools/time_ops.py bash tools/ab_time.sh [args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in 1 2; do
    for so in ab/*.so; do
        echo "== $so"
        if [ -n "$TOOL" ]; then FLOW2D_HIP_LIB="$R/$so" timeout 300 python "$TOOL" "$@" 2>&1 | grep -E "us |ms"
        else FLOW2D_HIP_LIB="$R/$so" timeout 120 python tools/time_sweep.py ${@:-4096 4096 2} 2>&1 | grep -E "level solve"; fi
    done
done
