#!/bin/bash
# round 5, GPU call 48: bench lines of every workload with the new lock-step group sizes and the batch leg first; the driver's command
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh bench > gpurun_out/r05/measure_final3.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_final3.txt; exit 1; }
cat gpurun_out/r05/measure_final3.txt
python bench.py --workload cfg3_4096_sor > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05/sor_bench.err
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_driver_style_bench_line.json 2> gpurun_out/r05/driver_style.err; E=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("gpurun_out/r05_cfg3_4096_sor_bench_line.json")); print("sor", d["pairs_per_s"], d["pairs_per_s_single"], d["pairs_per_s_incl_h2d"], d["value"], d["roofline"]["avg_launch_ms"])
d=json.load(open("gpurun_out/r05_driver_style_bench_line.json")); print("driver cmd", d["pairs_per_s"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["valu_issue_frac"], "batch", d["batch"]["pairs_per_s"], "h2d", d["pairs_per_s_incl_h2d"], "ok", d["output_check"]["ok"], "wall", $E-$S)
PY
