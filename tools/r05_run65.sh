#!/bin/bash
# round 5, GPU call 65: last check of the tree as committed: GPU suite, smoke(), the driver's command
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_last.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_last.txt; exit 1; }
tail -n 2 gpurun_out/r05/gpu_tests_last.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/driver_cmd_last.json 2> gpurun_out/r05/driver_cmd_last.err; E=$(date +%s.%N)
python3 - <<PY
import json
d=json.load(open("gpurun_out/r05/driver_cmd_last.json"))
print("driver cmd", d["pairs_per_s"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], "batch", d["batch"]["pairs_per_s"], "h2d", d["pairs_per_s_incl_h2d"], "ok", d["output_check"]["ok"], "wall", $E-$S)
PY
