mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py tests/test_gpu_reference.py -x -q -k "solve_level or level or pyramid or compute_flow or rub or config or pipeline" > gpurun_out/r05/test_tiles.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_tiles.txt
tail -3 gpurun_out/r05/test_tiles.txt
WLS="cfg1_rub cfg2_1024_grey cfg3_4096_gradient" VAR=FLOW2D_TILE_OUTER VALUES="1 2" bash tools/env_ab.sh ab/dev.so > gpurun_out/r05/tile_two_outer_ab.txt 2>&1
cat gpurun_out/r05/tile_two_outer_ab.txt
