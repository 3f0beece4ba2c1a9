#!/bin/bash
# round 5, GPU call 66: the multi-rank branches of bench.py once more with the final defaults (4 ranks sharing the one GPU over gloo)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 600 python bench.py --gpus 4 --rehearse-on-one-gpu --steps 20 --warmup 5 > gpurun_out/r05/rehearse4.json 2> gpurun_out/r05/rehearse4.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r05/rehearse4.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d.get("rehearsal"), d["pairs_per_s"], "check", d["output_check"]["ok"], d["output_check"].get("oracle"), "batch", d["batch"]["pairs_per_s"], d["batch"]["gather"], "h2d", d["pairs_per_s_incl_h2d"], "mem", d["device_memory"]["used_gib"])
PY
