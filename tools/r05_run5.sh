set -x
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "streaming or sor" > gpurun_out/r05/test_sweeps2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sweeps2.txt
tail -3 gpurun_out/r05/test_sweeps2.txt
for rows in 16 32 64 128; do for s in 4096 8192; do echo "== rows $rows" >> gpurun_out/r05/time_per_sweep_rows.txt; FLOW2D_HIP_LIB=$PWD/ab/dev.so FLOW2D_SWEEP_ROWS=$rows timeout -k 10 120 python tools/time_per_sweep.py $s $s 2>&1 | grep sweep >> gpurun_out/r05/time_per_sweep_rows.txt; done; done
cat gpurun_out/r05/time_per_sweep_rows.txt
