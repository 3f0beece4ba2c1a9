mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py -x -q -k "sor" > gpurun_out/r05/test_sor2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sor2.txt
tail -n 4 gpurun_out/r05/test_sor2.txt
timeout -k 10 900 python tools/fuzz_parity.py 1500 201 0 0.35 > gpurun_out/r05/fuzz_auto_final.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_final.txt
timeout -k 10 900 python tools/fuzz_parity.py 1500 202 2 0.35 > gpurun_out/r05/fuzz_fused_final.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_fused_final.txt
timeout -k 10 600 python tools/fuzz_reference.py 1000 203 > gpurun_out/r05/fuzz_reference_final.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_final.txt
timeout -k 10 600 python bench.py --workload cfg3_4096_sor --no-pmc > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05_cfg3_4096_sor_bench.err; echo sor rc=$?
