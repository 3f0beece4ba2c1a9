mkdir -p gpurun_out/r05
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r05/test_gpu_all2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_gpu_all2.txt
tail -n 3 gpurun_out/r05/test_gpu_all2.txt
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/half_weights_ab.txt 2>&1
WLS="cfg3_4096_gradient cfg3_4096_grey" bash tools/ab_bench.sh ab/a_half_weights.so ab/b_full_weights.so >> gpurun_out/r05/half_weights_ab.txt 2>&1
cat gpurun_out/r05/half_weights_ab.txt
timeout -k 10 600 python tools/fuzz_parity.py 600 301 2 0.2 > gpurun_out/r05/fuzz_half_weights.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_half_weights.txt
timeout -k 10 600 python tools/fuzz_reference.py 400 302 >> gpurun_out/r05/fuzz_half_weights.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_half_weights.txt
