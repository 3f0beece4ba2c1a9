#!/bin/bash
# round 5, GPU call 46: the driver's command with the new lock-step group sizes: all legs, wall time; the other workloads' lines
set -e
mkdir -p gpurun_out/r05
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/driver_cmd_groups.json 2> gpurun_out/r05/driver_cmd_groups.err; E=$(date +%s.%N)
python3 - <<PY
import json
d=json.load(open("gpurun_out/r05/driver_cmd_groups.json"))
print("driver cmd", d["pairs_per_s"], d["value"], d["ms_per_step"], d["config"]["steps_per_lock_step_group"], "single", d["pairs_per_s_single"], "h2d", d["pairs_per_s_incl_h2d"], "batch", d["batch"]["pairs_per_s"], "check", d["output_check"]["ok"], d["output_check"].get("oracle"), "mem", d["device_memory"]["used_gib"], "wall", $E-$S)
print(json.dumps(d.get("host_entry"))[:600])
PY
for wl in cfg2_1024_grey cfg1_rub cfg5_8192_grey; do
python3 bench.py --workload $wl --no-pmc --no-batch-leg --no-reference-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', d['pairs_per_s'], d['ms_per_step'], 'group', d['config']['steps_per_lock_step_group'], 'single', d['pairs_per_s_single'], 'h2d', d['pairs_per_s_incl_h2d'], 'check', d['output_check']['ok'], 'mem', d['device_memory']['used_gib'])"
done
