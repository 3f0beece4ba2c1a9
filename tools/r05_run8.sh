set -x
mkdir -p gpurun_out/r05
timeout -k 10 900 python bench.py --workload cfg3_4096_sor --steps 20 --no-pmc > gpurun_out/r05/sor_line.json 2> gpurun_out/r05/sor_line.err; echo "sor bench rc=$?"
timeout -k 10 300 python -m pytest tests/test_gpu_reference.py -x -q -k "fma" -s > gpurun_out/r05/test_fma.txt 2>&1; echo "rc=$?"
cat gpurun_out/fma_contraction_rmse.json
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r05/bench_line_3.json 2> gpurun_out/r05/bench_line_3.err; echo "bench rc=$?"
