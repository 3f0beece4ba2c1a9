#!/bin/bash
# A/B of whole-pipeline rates on ONE GPU box: every library given (default: the in-tree one and ab/*.so) runs the bench's
# timed region for the workloads given in WLS, twice, without the side legs.  usage: [WLS="cfg3_4096_gradient ..."] bash tools/ab_bench.sh [libs...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
LIBS=${@:-cuda-flow2d_amd/csrc/libflow2d_hip.so ab/*.so}
WLS=${WLS:-cfg3_4096_gradient cfg3_4096_grey cfg2_1024_grey cfg4_1080p_batch cfg5_8192_grey}
for rep in 1 2; do
    for wl in $WLS; do
        for so in $LIBS; do
            FLOW2D_HIP_LIB="$R/$so" timeout -k 10 300 python3 bench.py --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg \
                --no-cpu-baseline --no-reference-baseline --no-batch-leg --no-probe-builds 2>/dev/null |
                python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s %-40s pairs/s %8.1f  ms/step %7.3f  launch_ms %s  single_pair_ms %s' % ('$wl', '$so', d['pairs_per_s'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d.get('single_pair_latency_ms')))"
        done
    done
done
