set -x
mkdir -p gpurun_out/r05
O=gpurun_out/r05/strips_per_column_ab.txt
for rep in 1 2; do
for ny in 0 25 28 31 34 37 43 50; do
  echo "== FLOW2D_FUSED_NY=$ny (lone 4096^2 level solve, 10 x 5)" >> $O
  FLOW2D_HIP_LIB=$PWD/ab/dev.so FLOW2D_FUSED_NY=$ny timeout -k 10 120 python tools/time_sweep.py 4096 4096 2 5 2>&1 | grep "level solve" >> $O
done
done
WLS="cfg3_4096_gradient" VAR=FLOW2D_FUSED_NY VALUES="0 28 31 37" bash tools/env_ab.sh ab/dev.so >> $O 2>&1
cat $O
timeout -k 10 900 python bench.py --workload cfg3_4096_sor --steps 20 --no-pmc > gpurun_out/r05/sor_line.json 2> gpurun_out/r05/sor_line.err; echo "sor bench rc=$?"
