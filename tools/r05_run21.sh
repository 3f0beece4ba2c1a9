#!/bin/bash
# round 5, GPU call 21: the multi-rank branches of bench.py walked on the one-GPU box (ranks share GPU 0, gloo)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 500 python bench.py --gpus 2 --rehearse-on-one-gpu > gpurun_out/r05/rehearse2.json 2> gpurun_out/r05/rehearse2.err
tail -c 600 gpurun_out/r05/rehearse2.json
timeout -k 10 300 python bench.py --gpus 3 --rehearse-on-one-gpu --workload cfg2_1024_grey --steps 12 > gpurun_out/r05/rehearse3.json 2> gpurun_out/r05/rehearse3.err
tail -c 300 gpurun_out/r05/rehearse3.json
