set -x
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference.py -x -q -k "streaming or sor or sweep or solver_kernels" > gpurun_out/r05/test_sweeps3.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sweeps3.txt
tail -4 gpurun_out/r05/test_sweeps3.txt
for s in 4096 8192; do timeout -k 10 200 python tools/time_per_sweep.py $s $s 2>&1 | grep -E "sweep|SOR" >> gpurun_out/r05/time_per_sweep_final.txt; done
cat gpurun_out/r05/time_per_sweep_final.txt
