#!/bin/bash
# round 5, GPU call 38: the library after the scalar trims: GPU suite, fuzzers, measurement pass
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_final2.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_final2.txt; exit 1; }
tail -n 2 gpurun_out/r05/gpu_tests_final2.txt
timeout -k 10 600 python tools/fuzz_parity.py 1500 701 0 0.3 > gpurun_out/r05/fuzz_auto_trim.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_trim.txt
timeout -k 10 600 python tools/fuzz_parity.py 1500 702 2 0.35 > gpurun_out/r05/fuzz_strips_trim.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_strips_trim.txt
timeout -k 10 400 python tools/fuzz_reference.py 600 703 > gpurun_out/r05/fuzz_reference_trim.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_trim.txt
ROUND=r05 bash tools/measure.sh bench > gpurun_out/r05/measure_final2.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_final2.txt; exit 1; }
cat gpurun_out/r05/measure_final2.txt
