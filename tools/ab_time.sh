#!/bin/bash
# A/B timing of solver builds on ONE GPU box (boxes differ by a few %, so variants must share a run):
# every ab/*.so is copied over csrc/libflow2d_hip.so in turn and timed with tools/time_sweep.py, twice.
# usage (on the GPU box): bash tools/ab_time.sh [time_sweep args]      or   TOOL=tools/time_ops.py bash tools/ab_time.sh [args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
cp cuda-flow2d_amd/csrc/libflow2d_hip.so /tmp/libflow2d_hip.keep
for rep in 1 2; do
    for so in ab/*.so; do
        cp "$so" cuda-flow2d_amd/csrc/libflow2d_hip.so
        echo "== $so"
        if [ -n "$TOOL" ]; then timeout 300 python "$TOOL" "$@" 2>&1 | grep -E "us |ms"
        else timeout 120 python tools/time_sweep.py ${@:-4096 4096 2} 2>&1 | grep -E "level solve"; fi
    done
done
cp /tmp/libflow2d_hip.keep cuda-flow2d_amd/csrc/libflow2d_hip.so
