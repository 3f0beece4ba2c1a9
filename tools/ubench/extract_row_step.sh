#!/bin/bash
# One steady-state row step of the strip kernel as the compiler emitted it, for tools/ubench/gen_replay.py:
#   bash tools/fused_isa_report.sh [flags]          -> "ISA kept in /tmp/fused_isa.XXXX"
#   bash tools/ubench/extract_row_step.sh /tmp/fused_isa.XXXX/solve_fused_g1.s LBB0_179 LBB0_183 > build_ubench/steps/name.s
# (the labels: two consecutive blocks of the interior loop in the report's list, "instr 4xx valu 3xx")
awk -v a="^\\\\." -v from="$2" -v to="$3" '$0 ~ "^\\."from":"{p=1} p && $0 ~ "^\\."to":"{exit} p' "$1"
