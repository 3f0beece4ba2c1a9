#!/bin/bash
# round 5, GPU call 32: the final library (issue-priority strip kernel; log-derivative instances packed): GPU suite, fuzzers,
# the SOR workload's line, the default command as the driver runs it (wall time)
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_final.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_final.txt; exit 1; }
tail -n 2 gpurun_out/r05/gpu_tests_final.txt
timeout -k 10 900 python tools/fuzz_parity.py 2500 501 0 0.3 > gpurun_out/r05/fuzz_auto_issue_priority.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_issue_priority.txt
timeout -k 10 900 python tools/fuzz_parity.py 1500 502 2 0.35 > gpurun_out/r05/fuzz_strips_issue_priority.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_strips_issue_priority.txt
timeout -k 10 600 python tools/fuzz_reference.py 800 503 > gpurun_out/r05/fuzz_reference_issue_priority.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_issue_priority.txt
python bench.py --workload cfg3_4096_sor > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05/sor_bench.err
S=$(date +%s.%N); python bench.py > gpurun_out/r05_driver_style_bench_line.json 2> gpurun_out/r05/driver_style.err; E=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("gpurun_out/r05_cfg3_4096_sor_bench_line.json")); print("sor", d["pairs_per_s"], d["pairs_per_s_single"], json.dumps(d.get("sor_time_to_residual"))[:1200])
d=json.load(open("gpurun_out/r05_driver_style_bench_line.json")); print("default", d["pairs_per_s"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], "batch", d["batch"]["pairs_per_s"], "wall", $E-$S)
PY
