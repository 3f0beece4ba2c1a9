#!/bin/bash
# Developer experiment: what does occupancy buy the fused strip kernel?  The inner = 2 instantiation needs 130-142
# VGPRs, so the SAME code runs at 1, 2 or 3 waves per SIMD depending only on how many workgroups a CU admits, which the
# dynamic-LDS pad knob (FLOW2D_FUSED_LDS_PAD) controls; ab/inner2_w4.so is the same source under __launch_bounds__(256, 4).
# usage (GPU box): bash tools/occupancy_exp.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
cp cuda-flow2d_amd/csrc/libflow2d_hip.so /tmp/libflow2d_hip.keep
run() { # so pad rows label
    cp "$1" cuda-flow2d_amd/csrc/libflow2d_hip.so
    echo "== $4: $1 pad=$2 rows=$3"
    FLOW2D_FUSED_LDS_PAD=$2 FLOW2D_FUSED_ROWS=$3 timeout -k 10 120 python tools/time_sweep.py 4096 4096 2 2 2>&1 | grep "level solve"
}
for rep in 1 2; do
    run ab/inner2_w2.so 81920 293 "1 wave/SIMD (1 WG/CU, 252 WGs)"
    run ab/inner2_w2.so 60000 147 "2 waves/SIMD (2 WG/CU, 504 WGs)"
    run ab/inner2_w2.so 0 98 "3 waves/SIMD (3 WG/CU, 756 WGs)"
    run ab/inner2_w4.so 0 74 "4 waves/SIMD (4 WG/CU, 1008 WGs; 128 VGPRs, spills)"
    # the same with every row address folded into the first 8 rows of the planes (cache-resident: compute only)
    run ab/inner2_nomem_w2.so 81920 293 "compute only, 1 wave/SIMD"
    run ab/inner2_nomem_w2.so 60000 147 "compute only, 2 waves/SIMD"
    run ab/inner2_nomem_w2.so 0 98 "compute only, 3 waves/SIMD"
    run ab/inner2_nomem_w4.so 0 74 "compute only, 4 waves/SIMD"
done
cp /tmp/libflow2d_hip.keep cuda-flow2d_amd/csrc/libflow2d_hip.so
