#!/bin/bash
# Counter passes (rocprofv3 --pmc, separate passes, kernel trace only) over any command, summarised for the kernels whose
# name contains PATTERN.  Run on the GPU box: bash tools/pmc_kernel.sh PATTERN python3 tools/run_x_levels.py 8192
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PATTERN=$1; shift
OUT=$R/gpurun_out/pmc_kernel
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/p$i" -- "$@" > "$OUT/p$i.log" 2>&1
    echo "pass $i ($c): rc $?"
    cp "$OUT"/p$i/*/*counter_collection.csv "$OUT/pass$i.csv" 2>/dev/null
    rm -rf "$OUT/p$i"
done
python3 - "$PATTERN" "$OUT"/pass*.csv <<'PY'
import csv, sys, collections
pat = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if pat not in r["Kernel_Name"]: continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        key = "%s grid %s wg %s vgpr %s lds %s" % (name, r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["LDS_Block_Size"])
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[key]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, cs in sorted(d.items()):
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 2:]
        print("   %-26s mean %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
