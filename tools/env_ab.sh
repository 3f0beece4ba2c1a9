#!/bin/bash
# Developer A/B (GPU box): one environment switch of a developer build (-DFLOW2D_DEV_BUILD) over the bench's timed region, e.g. the strip
# planner's work weight (FLOW2D_FUSED_PLAN_BIAS) or AUTO's tile / strip crossover (FLOW2D_TILED_MAX_PIXELS).
# usage: [WLS="..."] VAR=FLOW2D_FUSED_PLAN_BIAS VALUES="0 1 2" bash tools/env_ab.sh ab/dev.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
SO=${1:-ab/dev.so}
VAR=${VAR:-FLOW2D_FUSED_PLAN_BIAS}
WLS=${WLS:-cfg3_4096_gradient cfg2_1024_grey cfg4_1080p_batch}
VALUES=${VALUES:-0 1 2 4}
for rep in 1 2; do
    for wl in $WLS; do
        for b in $VALUES; do
            env $VAR=$b FLOW2D_HIP_LIB="$R/$SO" timeout -k 10 300 python3 bench.py --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg \
                --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null |
                python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s $VAR=%-8s pairs/s %8.1f  ms/step %7.3f  launch_ms %s  single_pair_ms %s' % ('$wl', '$b', d['pairs_per_s'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d.get('single_pair_latency_ms')))"
        done
    done
done
