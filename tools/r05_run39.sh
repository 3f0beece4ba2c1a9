#!/bin/bash
# round 5, GPU call 39: kernel traces of the final library (profiles/r05_*_by_grid.txt), the SOR and driver-style lines
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh trace trace_default > gpurun_out/r05/measure_traces_final.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_traces_final.txt; exit 1; }
grep "fused_outer_kernel<5, [01], true, false, false>  *131072\|^kernel" gpurun_out/r05/measure_traces_final.txt | head
python bench.py --workload cfg3_4096_sor > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05/sor_bench.err
S=$(date +%s.%N); python bench.py > gpurun_out/r05_driver_style_bench_line.json 2> gpurun_out/r05/driver_style.err; E=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("gpurun_out/r05_cfg3_4096_sor_bench_line.json")); print("sor", d["pairs_per_s"], d["pairs_per_s_single"], d["pairs_per_s_incl_h2d"], d["value"], d["roofline"]["avg_launch_ms"])
d=json.load(open("gpurun_out/r05_driver_style_bench_line.json")); print("default", d["pairs_per_s"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["valu_issue_frac"], "batch", d["batch"]["pairs_per_s"], "wall", $E-$S)
PY
