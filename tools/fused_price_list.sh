#!/bin/bash
# Developer probe: what one more instruction of a class costs in the fused kernel's own stream.  Builds six developer libraries --
# FLOW2D_FUSED_INJECT=1..5 put 24 extra independent instructions of one class (plain v_add_f32, v_pk_add_f32, v_mov_b32_dpp,
# v_rcp_f32, s_nop) into every steady-state row step (four places of six), 0 is the same build without them -- into ab/inj*.so;
# time them on the GPU box with  bash tools/ab_time.sh 4096 4096 2  (level solve = ten launches; profiles/r04_experiments/fused_price_list.txt).
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R/cuda-flow2d_amd/csrc" || exit 1
for k in 0 1 2 3 4 5; do
    extra="-DFLOW2D_DEV_BUILD -DFLOW2D_FUSED_DEV"
    [ $k != 0 ] && extra="$extra -DFLOW2D_FUSED_INJECT=$k"
    make -s -j8 BUILD=build_inj$k LIB="$R/ab/inj$k.so" EXTRA="$extra" || exit 1
done
