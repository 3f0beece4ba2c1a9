#!/bin/bash
# round 5, GPU call 67: the round's final measurement pass, everything on ONE box: bench lines, eager traces, the default command's
# trace, the SOR line, the driver's command
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh bench trace trace_default > gpurun_out/r05/measure_final_pass.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_final_pass.txt; exit 1; }
grep "^cfg\|fused_outer_kernel<5, [01], true, false, false>  *131072" gpurun_out/r05/measure_final_pass.txt
python bench.py --workload cfg3_4096_sor > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05/sor_bench.err
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_driver_style_bench_line.json 2> gpurun_out/r05/driver_style.err; E=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("gpurun_out/r05_cfg3_4096_sor_bench_line.json")); print("sor", d["pairs_per_s"], d["pairs_per_s_single"], d["pairs_per_s_incl_h2d"], d["value"], d["roofline"]["avg_launch_ms"])
d=json.load(open("gpurun_out/r05_driver_style_bench_line.json")); print("driver cmd", d["pairs_per_s"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["valu_issue_frac"], "batch", d["batch"]["pairs_per_s"], "h2d", d["pairs_per_s_incl_h2d"], "ok", d["output_check"]["ok"], "wall", $E-$S)
PY
