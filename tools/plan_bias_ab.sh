#!/bin/bash
# Developer A/B (GPU box): the strip planner's throughput bias (FLOW2D_FUSED_PLAN_BIAS, developer builds only) over the
# bench's timed region.  usage: [WLS="..."] [BIASES="0 0.5 1 2"] bash tools/plan_bias_ab.sh ab/bias.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
SO=${1:-ab/bias.so}
WLS=${WLS:-cfg3_4096_gradient cfg2_1024_grey cfg4_1080p_batch}
BIASES=${BIASES:-0 0.5 1 2 4}
for rep in 1 2; do
    for wl in $WLS; do
        for b in $BIASES; do
            FLOW2D_FUSED_PLAN_BIAS=$b FLOW2D_HIP_LIB="$R/$SO" timeout -k 10 300 python3 bench.py --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg \
                --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null |
                python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s bias %-4s pairs/s %8.1f  ms/step %7.3f  launch_ms %s  single_pair_ms %s' % ('$wl', '$b', d['pairs_per_s'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d.get('single_pair_latency_ms')))"
        done
    done
done
