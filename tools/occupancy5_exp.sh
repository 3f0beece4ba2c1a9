#!/bin/bash
# Developer experiment (round 5): what would a THIRD wave per SIMD buy the five-sweep strip kernel?  The product kernel needs
# 213-215 registers (two waves per SIMD); the probe build -DFLOW2D_FUSED_SHORT_RING=3 -DFLOW2D_FUSED_WAVES=3 runs the SAME
# instruction stream per row step with a coefficient ring of three entries and one input row in flight (WRONG results, timing
# only) in 156-158 registers, so one binary runs at 1, 2 or 3 waves per SIMD depending on the dynamic-LDS pad
# (FLOW2D_FUSED_LDS_PAD); strip heights follow so that every case is one round of waves.
# build (here):  D="-DFLOW2D_DEV_BUILD -DFLOW2D_FUSED_DEV"
#   make -C cuda-flow2d_amd/csrc BUILD=build_dev LIB=$PWD/ab/dev.so EXTRA="$D"
#   make -C cuda-flow2d_amd/csrc BUILD=build_short3 LIB=$PWD/ab/short3.so EXTRA="$D -DFLOW2D_FUSED_SHORT_RING=3 -DFLOW2D_FUSED_WAVES=3"
#   ... short3_nomem.so with -DFLOW2D_FUSED_COMPUTE_ONLY, short3_mem.so with -DFLOW2D_FUSED_MEMORY_ONLY added
# usage (GPU box): bash tools/occupancy5_exp.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() { # so pad rows label
    echo "== $4: $1 pad=$2 rows=$3"
    FLOW2D_HIP_LIB="$R/$1" FLOW2D_FUSED_LDS_PAD=$2 FLOW2D_FUSED_ROWS=$3 timeout -k 10 120 python tools/time_sweep.py 4096 4096 2 5 2>&1 | grep "level solve"
}
for rep in 1 2; do
    run ab/dev.so 0 0 "product kernel, planner's strips (2 waves/SIMD)"
    run ab/dev.so 0 164 "product kernel, uniform strips of 164 rows (2 waves/SIMD, 500 WGs)"
    run ab/dev.so 81920 342 "product kernel, 1 wave/SIMD (240 WGs)"
    run ab/short3.so 81920 342 "probe, 1 wave/SIMD (1 WG/CU, 240 WGs)"
    run ab/short3.so 60000 164 "probe, 2 waves/SIMD (2 WG/CU, 500 WGs)"
    run ab/short3.so 0 108 "probe, 3 waves/SIMD (3 WG/CU, 760 WGs)"
    run ab/short3_nomem.so 81920 342 "probe, compute only, 1 wave/SIMD"
    run ab/short3_nomem.so 60000 164 "probe, compute only, 2 waves/SIMD"
    run ab/short3_nomem.so 0 108 "probe, compute only, 3 waves/SIMD"
    run ab/short3_mem.so 60000 164 "probe, loads and stores only, 2 waves/SIMD"
    run ab/short3_mem.so 0 108 "probe, loads and stores only, 3 waves/SIMD"
done
