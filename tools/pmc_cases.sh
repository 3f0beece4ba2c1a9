#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, separate passes, kernel trace only) over tools/pitch_ab.py --once for the cases given:
# where the fused kernel's wave cycles go, per launch geometry.  Run on the GPU box: bash tools/pmc_cases.sh "4096x4096" "1920x1080*8"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT/pmc_cases"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" \
         "SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_CYCLES" \
         "GRBM_GUI_ACTIVE TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    rm -rf "$OUT/pmc_cases/p$i"
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/pmc_cases/p$i" -- python3 "$R/tools/pitch_ab.py" --once --smooth "$@" > "$OUT/pmc_cases/p$i.log" 2>&1
    echo "pass $i ($c): rc $?"
    cp "$OUT"/pmc_cases/p$i/*/*counter_collection.csv "$OUT/pmc_cases/pass$i.csv" 2>/dev/null
    rm -rf "$OUT/pmc_cases/p$i"
done
python3 "$R/tools/pmc_cases_summary.py" "$OUT"/pmc_cases/pass*.csv
