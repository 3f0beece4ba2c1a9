#!/bin/bash
# A/B of whole-bench throughput on ONE GPU box: every ab/*.so in turn as csrc/libflow2d_hip.so, twice.
# usage (GPU box): bash tools/ab_bench.sh [bench.py args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
cp cuda-flow2d_amd/csrc/libflow2d_hip.so /tmp/libflow2d_hip.keep
for rep in 1 2; do
    for so in ab/*.so; do
        cp "$so" cuda-flow2d_amd/csrc/libflow2d_hip.so
        timeout -k 10 200 python3 bench.py --no-batch-leg --no-cpu-baseline --no-reference-baseline "$@" > gpurun_out/ab_bench.json 2> gpurun_out/ab_bench.err || tail -2 gpurun_out/ab_bench.err
        python3 -c "
import json; d=json.load(open('gpurun_out/ab_bench.json')); print('$so', d['pairs_per_s'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
    done
done
cp /tmp/libflow2d_hip.keep cuda-flow2d_amd/csrc/libflow2d_hip.so
