mkdir -p gpurun_out/r05
O=gpurun_out/r05/three_waves_taking_turns.txt
run() { echo "== $4: $1 pad=$2 rows=$3" >> $O; FLOW2D_HIP_LIB="$PWD/$1" FLOW2D_FUSED_LDS_PAD=$2 FLOW2D_FUSED_ROWS=$3 timeout -k 10 120 python tools/time_sweep.py 4096 4096 2 5 2>&1 | grep "level solve" >> $O; }
for rep in 1 2; do
  run ab/short3.so 60000 164 "probe, 2 waves/SIMD"
  run ab/short3.so 0 108 "probe, 3 waves/SIMD"
  run ab/short3_t11.so 0 108 "probe, 3 waves/SIMD, favoured slot rotates every 2^11 cycles"
  run ab/short3_t13.so 0 108 "probe, 3 waves/SIMD, every 2^13"
  run ab/short3_t15.so 0 108 "probe, 3 waves/SIMD, every 2^15"
  run ab/short3_t15.so 60000 164 "probe, 2 waves/SIMD, every 2^15 (three-way rotation on two slots)"
done
cat $O
