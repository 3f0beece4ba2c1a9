set -x
mkdir -p gpurun_out/r05
timeout -k 5 120 ./build_ubench/pair_latency > gpurun_out/r05/pair_latency.txt 2>&1 || exit 1
S() { # lib pad rows label
  echo "#### $4" >> gpurun_out/r05/stamps_occupancy.txt
  FLOW2D_HIP_LIB=$PWD/$1 FLOW2D_FUSED_LDS_PAD=$2 FLOW2D_FUSED_ROWS=$3 timeout -k 10 120 python tools/fused_wave_stamps.py 4096x4096 grad >> gpurun_out/r05/stamps_occupancy.txt 2>&1
}
S ab/stamps.so 0 0 "product kernel, planner's strips, 2 waves/SIMD" || exit 1
S ab/stamps.so 81920 342 "product kernel, 1 wave/SIMD, uniform strips of 342 rows" || exit 1
S ab/stamps_short3.so 81920 342 "probe, 1 wave/SIMD" || exit 1
S ab/stamps_short3.so 60000 164 "probe, 2 waves/SIMD" || exit 1
S ab/stamps_short3.so 0 108 "probe, 3 waves/SIMD" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_batch_tool.py -x -q > gpurun_out/r05/test_batch_tool.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_batch_tool.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r05/bench_line_1.json 2> gpurun_out/r05/bench_line_1.err; echo "bench rc=$?"
