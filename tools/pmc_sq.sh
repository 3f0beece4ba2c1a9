#!/bin/bash
# Two rocprofv3 --pmc passes with SQ counters over tools/pmc_workload.py: where the fused kernel's wave cycles go
# (issuing VALU / scalar, waiting on counters, stalled at issue).  Run on the GPU box: bash tools/pmc_sq.sh [size]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SIZE=${1:-4096}
OUT=$R/gpurun_out
mkdir -p "$OUT/pmc_csv"
cd /tmp && export TMPDIR=/tmp
i=4
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
         "SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_CYCLES"; do
    i=$((i + 1))
    rm -rf "$OUT/pmc_$i"
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/pmc_$i" -- python3 "$R/tools/pmc_workload.py" $SIZE > "$OUT/pmc_$i.log" 2>&1
    echo "pass $i ($c): rc $?"
    cp "$OUT"/pmc_$i/*/*counter_collection.csv "$OUT/pmc_csv/pass$i.csv" 2>/dev/null
done
python3 - "$OUT/pmc_csv/pass5.csv" "$OUT/pmc_csv/pass6.csv" <<'PY'
import csv, sys, collections
for path in sys.argv[1:]:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "fused_outer" not in r["Kernel_Name"]: continue
        key = r["Kernel_Name"].split("(")[0][-40:]
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[key]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, cs in d.items():
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-28s last %.4g  (n=%d)" % (c, v[-1], len(v)))
PY
