#!/bin/bash
# round 5, GPU call 26: the product library with the plain-instruction strip kernel + issue priority: GPU suite, bench lines
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_prio.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_prio.txt; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_prio.txt
python bench.py > gpurun_out/r05/bench_prio_default.json 2> gpurun_out/r05/bench_prio_default.err
python bench.py --workload cfg3_4096_grey --no-pmc --no-cpu-baseline --no-reference-baseline > gpurun_out/r05/bench_prio_grey.json 2>/dev/null
python bench.py --workload cfg2_1024_grey --no-pmc --no-cpu-baseline --no-reference-baseline --no-batch-leg > gpurun_out/r05/bench_prio_cfg2.json 2>/dev/null
python bench.py --workload cfg5_8192_grey --no-pmc --no-cpu-baseline --no-reference-baseline --no-batch-leg > gpurun_out/r05/bench_prio_cfg5.json 2>/dev/null
python - <<'PY'
import json
for n in ("default", "grey", "cfg2", "cfg5"):
    d = json.loads(open("gpurun_out/r05/bench_prio_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["config"]["workload"], d["value"], d.get("pairs_per_s"), d["ms_per_step"], "roofline", d["roofline"]["achieved"], d["roofline"].get("launch_ms"), "batch", (d.get("batch") or {}).get("pairs_per_s"), "ok", d["output_check"]["ok"])
PY
