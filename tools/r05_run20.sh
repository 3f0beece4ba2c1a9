mkdir -p gpurun_out/r05
( time python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r05/bench_driver_style.json 2> gpurun_out/r05/bench_driver_style.err ) 2> gpurun_out/r05/bench_driver_style.time; cat gpurun_out/r05/bench_driver_style.time
python3 -c "
import json; d=json.load(open('gpurun_out/r05/bench_driver_style.json')); print(d['pairs_per_s'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['batch']['pairs_per_s'], d['output_check']['ok'])"
bash tools/run_batch8.sh --world 1 --pairs 8 --repeat 16 > gpurun_out/r05/run_batch8_world1.txt 2>&1; echo "run_batch8 rc=$?"; tail -n 2 gpurun_out/r05/run_batch8_world1.txt | cut -c1-400
timeout -k 10 900 python tools/fuzz_parity.py 4000 401 0 0.3 > gpurun_out/r05/fuzz_big_auto.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_big_auto.txt
