set -x
mkdir -p gpurun_out/r05
timeout -k 5 120 ./build_ubench/regbank > gpurun_out/r05/regbank.txt 2>&1 || exit 1
timeout -k 5 120 ./build_ubench/stream10 > gpurun_out/r05/stream10_v2.txt 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "sor" > gpurun_out/r05/test_sor.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sor.txt
tail -5 gpurun_out/r05/test_sor.txt
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05/test_gpu_all.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_gpu_all.txt
tail -5 gpurun_out/r05/test_gpu_all.txt
timeout -k 10 300 python bench.py --workload cfg4_1080p_batch --steps 64 --no-pmc --no-cpu-baseline --no-reference-baseline --no-oracle-check --no-host-entry-leg > gpurun_out/r05/cfg4_64.json 2> /dev/null; echo "rc=$?"
timeout -k 10 300 python bench.py --workload cfg4_1080p_batch --steps 100 --no-pmc --no-cpu-baseline --no-reference-baseline --no-oracle-check --no-host-entry-leg > gpurun_out/r05/cfg4_100.json 2> /dev/null; echo "rc=$?"
