#!/bin/bash
# round 5, GPU call 49: config 4's line again (its host-entry leg had been given a batch object of single pairs)
set -e
mkdir -p gpurun_out/r05
python3 bench.py --workload cfg4_1080p_batch --no-batch-leg > gpurun_out/r05_cfg4_1080p_batch_bench_line.json 2> gpurun_out/r05/cfg4.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05_cfg4_1080p_batch_bench_line.json'))
print('cfg4', d['pairs_per_s'], d['ms_per_step_min'], d['ms_per_step_max'], 'single', d['pairs_per_s_single'], 'h2d', d['pairs_per_s_incl_h2d'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['effective_frac'], d['roofline']['valu_issue_frac'], d['output_check']['ok'])"
