#!/bin/bash
# Developer experiment: what does occupancy buy the fused strip kernel's instruction stream?  The inner = 2 instantiation needs 138-141
# VGPRs, so the SAME code runs at 1, 2 or 3 waves per SIMD depending only on how many workgroups a CU admits, which the dynamic-LDS
# pad of developer builds (FLOW2D_FUSED_LDS_PAD) controls; strip heights follow so that every case fills its wave slots once.
# build:  make -C cuda-flow2d_amd/csrc BUILD=build_dev LIB=$PWD/ab/dev.so EXTRA="-DFLOW2D_DEV_BUILD -DFLOW2D_FUSED_DEV"
#         (and ab/dev_nomem.so with -DFLOW2D_FUSED_COMPUTE_ONLY added: every row folded onto eight cache-resident rows)
# usage (GPU box): bash tools/occupancy_exp.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() { # so pad rows label
    echo "== $4: $1 pad=$2 rows=$3"
    FLOW2D_HIP_LIB="$R/$1" FLOW2D_FUSED_LDS_PAD=$2 FLOW2D_FUSED_ROWS=$3 timeout -k 10 120 python tools/time_sweep.py 4096 4096 2 2 2>&1 | grep "level solve"
}
for rep in 1 2; do
    run ab/dev.so 81920 293 "1 wave/SIMD (1 WG/CU, 252 WGs)"
    run ab/dev.so 60000 147 "2 waves/SIMD (2 WG/CU, 504 WGs)"
    run ab/dev.so 0 98 "3 waves/SIMD (3 WG/CU, 756 WGs)"
    run ab/dev_nomem.so 81920 293 "compute only, 1 wave/SIMD"
    run ab/dev_nomem.so 60000 147 "compute only, 2 waves/SIMD"
    run ab/dev_nomem.so 0 98 "compute only, 3 waves/SIMD"
done
