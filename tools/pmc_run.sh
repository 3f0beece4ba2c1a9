#!/bin/bash
# Collects SQ counters of the solver kernels over tools/time_sweep.py, one rocprofv3 --pmc pass per group
# (developer tool; run on the GPU box: bash tools/pmc_run.sh <tag>).  Summaries: tools/pmc_summary.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-pmc}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
         "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
         "SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_TRANS_F32" \
         "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i + 1))
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/g$i" -- python3 "$R/tools/time_sweep.py" 4096 4096 2 > "$OUT/g$i.log" 2>&1
    tail -1 "$OUT/g$i.log"
done
python3 "$R/tools/pmc_summary.py" "$OUT" fused
