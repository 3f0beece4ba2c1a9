set -x
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "sor or streaming or sweep" > gpurun_out/r05/test_sweeps.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sweeps.txt
tail -3 gpurun_out/r05/test_sweeps.txt
for s in 4096 8192; do timeout -k 10 120 python tools/time_per_sweep.py $s $s >> gpurun_out/r05/time_per_sweep.txt 2>&1; done
timeout -k 10 900 python bench.py --workload cfg3_4096_sor --steps 20 --no-pmc > gpurun_out/r05/sor_line.json 2> gpurun_out/r05/sor_line.err; echo "sor bench rc=$?"
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r05/bench_line_2.json 2> gpurun_out/r05/bench_line_2.err; echo "bench rc=$?"
