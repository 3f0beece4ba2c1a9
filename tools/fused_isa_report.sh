#!/bin/bash
# Developer report on the fused kernel's code: compiles csrc/solve_fused_instance.hip for Grey and Gradient with power-of-two
# spacings (dev instantiations only: inner 5) with the given extra flags, keeps the ISA and prints, per kernel, registers /
# scratch and -- for the blocks of the steady-state row loops -- instructions, scratch accesses (spill traffic inside the
# loop), LDS accesses, divisions.
# usage: bash tools/fused_isa_report.sh [extra hipcc flags...]
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/fused_isa.XXXX)
for k in 0 1; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-slp-vectorize \
        -I"$R/include" -DFLOW2D_DEV_BUILD -DFLOW2D_FUSED_DEV -DFLOW2D_FUSED_INSTANCE_GRAD=$k -DFLOW2D_FUSED_INSTANCE_POW2=1 "$@" -save-temps=obj \
        -Rpass-analysis=kernel-resource-usage -c "$R/cuda-flow2d_amd/csrc/solve_fused_instance.hip" -o "$T/solve_fused_g$k.o" 2>&1 |
        grep "Name\|VGPRs\|Scratch\|SGPRs Spill" | sed 's/^.*remark: [^ ]* *//; s/ \[-Rpass.*//'
    S=$T/solve_fused_g$k.s  # (-save-temps names its files after the source: keep each data term's ISA under a name of its own)
    mv "$T/solve_fused_instance-hip-amdgcn-amd-amdhsa-gfx950.s" "$S" || exit 1
    echo "== fused_outer_kernel<5, $k, true, false>: blocks inside loops"
    awk -v pat="fused_outer_kernelILi5ELi${k}ELb1ELb0" '$0 ~ "^_ZN.*"pat {p=1} p&&/^\.Lfunc_end/{exit} p' "$S" |
    awk '/^\.LBB|^; %bb\./{lbl=($1==";" ? $2 : $1); getline nxt; inloop[lbl]=(($0 nxt) ~ /in Loop|Loop Header/); $0=nxt} /scratch_/{c[lbl]++} /v_div_scale/{d[lbl]++} /ds_read|ds_write/{l[lbl]++} /^\tv_/{v[lbl]++} {n[lbl]++}
         END{for(k in n) if (inloop[k] && n[k] > 100) printf "%-10s instr %4d valu %4d scratch %2d lds %2d div_scale %2d\n", k, n[k], v[k]+0, c[k]+0, l[k]+0, d[k]+0}' | sort -t_ -k2 -n
done
echo "ISA kept in $T"
