#!/bin/bash
# A/B on ONE GPU box: for every ab/*.so, quick parity of the fused solver against the oracle, then the 4096^2 level-solve
# time (tools/time_sweep.py), twice.   usage (GPU box): bash tools/ab_parity_time.sh [time_sweep args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
cp cuda-flow2d_amd/csrc/libflow2d_hip.so /tmp/libflow2d_hip.keep
for so in ab/*.so; do
    cp "$so" cuda-flow2d_amd/csrc/libflow2d_hip.so
    echo "== $so"
    timeout -k 10 200 python tools/parity_quick.py 2>&1 | tail -5
done
for rep in 1 2; do
    for so in ab/*.so; do
        cp "$so" cuda-flow2d_amd/csrc/libflow2d_hip.so
        echo "== $so"
        timeout -k 10 120 python tools/time_sweep.py ${@:-4096 4096 2} 2>&1 | grep -E "level solve"
    done
done
cp /tmp/libflow2d_hip.keep cuda-flow2d_amd/csrc/libflow2d_hip.so
