#!/bin/bash
# The four rocprofv3 --pmc passes behind profiles/traffic.json (run on the GPU box: bash tools/pmc_passes.sh [size]).
# Counters go in separate passes, each with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SIZE=${1:-4096}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVES"; do
    i=$((i + 1))
    rm -rf "$OUT/pmc_$i"
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/pmc_$i" -- python3 "$R/tools/pmc_workload.py" $SIZE > "$OUT/pmc_$i.log" 2>&1
    echo "pass $i ($c): rc $?"
done
python3 "$R/tools/pmc_traffic.py" "$OUT" $SIZE > "$OUT/traffic_summary.json"
mkdir -p "$OUT/pmc_csv"
for i in 1 2 3 4; do cp "$OUT"/pmc_$i/*/*counter_collection.csv "$OUT/pmc_csv/pass$i.csv" 2>/dev/null; done
