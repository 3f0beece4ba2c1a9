cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo rc=$?; grep -E "passed|failed" gpurun_out/gpu_tests.log | tail -2
python3 bench.py --steps 100 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -2 gpurun_out/bench_default.err
python3 -c "
import json; d=json.load(open('gpurun_out/bench_default.json'))
print(d['value'], d['pairs_per_s'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['batch']['pairs_per_s'], d['output_check']['ok'], d['reference_gpu_baseline']['pairs_per_s'])"
for wl in cfg2_1024_grey cfg3_4096_grey; do python3 bench.py --workload $wl --steps 100 --no-batch-leg > gpurun_out/bench_$wl.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$wl.json')); print('$wl', d['value'], d['pairs_per_s'], d['ms_per_step'], d['output_check']['ok'])"; done
