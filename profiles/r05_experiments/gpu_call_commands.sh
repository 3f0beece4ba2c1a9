# Round 5's GPU calls, one section per call (formerly tools/r05_runNN.sh): the literal command lists, kept so that a figure in
# profiles/r05_experiments/ can be traced to the command that produced it.  Not meant to be run as a whole: copy a section.
# (Sections that name switches of the strip kernel removed in round 6 -- see csrc/solve_fused_probes.hpp -- need the tree of that round.)

####################################################################################################
# [2] r05_run2.sh
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

####################################################################################################
# [3] r05_run3.sh
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

####################################################################################################
# [4] r05_run4.sh
set -x
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "sor or streaming or sweep" > gpurun_out/r05/test_sweeps.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sweeps.txt
tail -3 gpurun_out/r05/test_sweeps.txt
for s in 4096 8192; do timeout -k 10 120 python tools/time_per_sweep.py $s $s >> gpurun_out/r05/time_per_sweep.txt 2>&1; done
timeout -k 10 900 python bench.py --workload cfg3_4096_sor --steps 20 --no-pmc > gpurun_out/r05/sor_line.json 2> gpurun_out/r05/sor_line.err; echo "sor bench rc=$?"
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r05/bench_line_2.json 2> gpurun_out/r05/bench_line_2.err; echo "bench rc=$?"

####################################################################################################
# [5] r05_run5.sh
set -x
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "streaming or sor" > gpurun_out/r05/test_sweeps2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sweeps2.txt
tail -3 gpurun_out/r05/test_sweeps2.txt
for rows in 16 32 64 128; do for s in 4096 8192; do echo "== rows $rows" >> gpurun_out/r05/time_per_sweep_rows.txt; FLOW2D_HIP_LIB=$PWD/ab/dev.so FLOW2D_SWEEP_ROWS=$rows timeout -k 10 120 python tools/time_per_sweep.py $s $s 2>&1 | grep sweep >> gpurun_out/r05/time_per_sweep_rows.txt; done; done
cat gpurun_out/r05/time_per_sweep_rows.txt

####################################################################################################
# [6] r05_run6.sh
set -x
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference.py -x -q -k "streaming or sor or sweep or solver_kernels" > gpurun_out/r05/test_sweeps3.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sweeps3.txt
tail -4 gpurun_out/r05/test_sweeps3.txt
for s in 4096 8192; do timeout -k 10 200 python tools/time_per_sweep.py $s $s 2>&1 | grep -E "sweep|SOR" >> gpurun_out/r05/time_per_sweep_final.txt; done
cat gpurun_out/r05/time_per_sweep_final.txt

####################################################################################################
# [7] r05_run7.sh
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

####################################################################################################
# [8] r05_run8.sh
set -x
mkdir -p gpurun_out/r05
timeout -k 10 900 python bench.py --workload cfg3_4096_sor --steps 20 --no-pmc > gpurun_out/r05/sor_line.json 2> gpurun_out/r05/sor_line.err; echo "sor bench rc=$?"
timeout -k 10 300 python -m pytest tests/test_gpu_reference.py -x -q -k "fma" -s > gpurun_out/r05/test_fma.txt 2>&1; echo "rc=$?"
cat gpurun_out/fma_contraction_rmse.json
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r05/bench_line_3.json 2> gpurun_out/r05/bench_line_3.err; echo "bench rc=$?"

####################################################################################################
# [10] r05_run10.sh
mkdir -p gpurun_out/r05
B() { timeout -k 10 400 python bench.py --steps 20 --no-cpu-baseline --no-reference-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-60s main %.1f pairs/s  batch leg %.1f pairs/s' % ('$*', d['pairs_per_s'], d['batch']['pairs_per_s']))"; }
B --no-pmc
B --no-pmc --no-host-entry-leg
B --no-pmc --no-oracle-check
B --no-pmc --no-host-entry-leg --no-oracle-check
B

####################################################################################################
# [12] r05_run12.sh
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py tests/test_gpu_reference.py -x -q -k "solve_level or level or pyramid or compute_flow or rub or config or pipeline" > gpurun_out/r05/test_tiles.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_tiles.txt
tail -3 gpurun_out/r05/test_tiles.txt
WLS="cfg1_rub cfg2_1024_grey cfg3_4096_gradient" VAR=FLOW2D_TILE_OUTER VALUES="1 2" bash tools/env_ab.sh ab/dev.so > gpurun_out/r05/tile_two_outer_ab.txt 2>&1
cat gpurun_out/r05/tile_two_outer_ab.txt

####################################################################################################
# [17] r05_run17.sh
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py -x -q -k "sor" > gpurun_out/r05/test_sor2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_sor2.txt
tail -n 4 gpurun_out/r05/test_sor2.txt
timeout -k 10 900 python tools/fuzz_parity.py 1500 201 0 0.35 > gpurun_out/r05/fuzz_auto_final.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_final.txt
timeout -k 10 900 python tools/fuzz_parity.py 1500 202 2 0.35 > gpurun_out/r05/fuzz_fused_final.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_fused_final.txt
timeout -k 10 600 python tools/fuzz_reference.py 1000 203 > gpurun_out/r05/fuzz_reference_final.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_final.txt
timeout -k 10 600 python bench.py --workload cfg3_4096_sor --no-pmc > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05_cfg3_4096_sor_bench.err; echo sor rc=$?

####################################################################################################
# [18] r05_run18.sh
mkdir -p gpurun_out/r05
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r05/test_gpu_all2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05/test_gpu_all2.txt
tail -n 3 gpurun_out/r05/test_gpu_all2.txt
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/half_weights_ab.txt 2>&1
WLS="cfg3_4096_gradient cfg3_4096_grey" bash tools/ab_bench.sh ab/a_half_weights.so ab/b_full_weights.so >> gpurun_out/r05/half_weights_ab.txt 2>&1
cat gpurun_out/r05/half_weights_ab.txt
timeout -k 10 600 python tools/fuzz_parity.py 600 301 2 0.2 > gpurun_out/r05/fuzz_half_weights.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_half_weights.txt
timeout -k 10 600 python tools/fuzz_reference.py 400 302 >> gpurun_out/r05/fuzz_half_weights.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_half_weights.txt

####################################################################################################
# [19] r05_run19.sh
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

####################################################################################################
# [20] r05_run20.sh
mkdir -p gpurun_out/r05
( time python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r05/bench_driver_style.json 2> gpurun_out/r05/bench_driver_style.err ) 2> gpurun_out/r05/bench_driver_style.time; cat gpurun_out/r05/bench_driver_style.time
python3 -c "
import json; d=json.load(open('gpurun_out/r05/bench_driver_style.json')); print(d['pairs_per_s'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['batch']['pairs_per_s'], d['output_check']['ok'])"
bash tools/run_batch8.sh --world 1 --pairs 8 --repeat 16 > gpurun_out/r05/run_batch8_world1.txt 2>&1; echo "run_batch8 rc=$?"; tail -n 2 gpurun_out/r05/run_batch8_world1.txt | cut -c1-400
timeout -k 10 900 python tools/fuzz_parity.py 4000 401 0 0.3 > gpurun_out/r05/fuzz_big_auto.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_big_auto.txt

####################################################################################################
# [21] r05_run21.sh
# round 5, GPU call 21: the multi-rank branches of bench.py walked on the one-GPU box (ranks share GPU 0, gloo)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 500 python bench.py --gpus 2 --rehearse-on-one-gpu > gpurun_out/r05/rehearse2.json 2> gpurun_out/r05/rehearse2.err
tail -c 600 gpurun_out/r05/rehearse2.json
timeout -k 10 300 python bench.py --gpus 3 --rehearse-on-one-gpu --workload cfg2_1024_grey --steps 12 > gpurun_out/r05/rehearse3.json 2> gpurun_out/r05/rehearse3.err
tail -c 300 gpurun_out/r05/rehearse3.json

####################################################################################################
# [22] r05_run22.sh
# round 5, GPU call 22: share and clustering of special instructions against the SIMD's issue rate (tools/ubench/gen_issue_mix.py)
set -e
mkdir -p gpurun_out/r05
timeout -k 5 240 ./build_ubench/issue_mix 3000 > gpurun_out/r05/issue_mix.txt
tail -n 5 gpurun_out/r05/issue_mix.txt

####################################################################################################
# [23] r05_run23.sh
# round 5, GPU call 23: the strip kernel's row step without packed arithmetic and / or without lane shifts (timing probes)
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/plain_stream_probe.txt 2>&1
cat gpurun_out/r05/plain_stream_probe.txt

####################################################################################################
# [24] r05_run24.sh
# round 5, GPU call 24: the strip kernel's own row-step instruction streams replayed (tools/ubench/gen_replay.py)
set -e
mkdir -p gpurun_out/r05
timeout -k 5 120 ./build_ubench/replay 2000 > gpurun_out/r05/replay.txt
cat gpurun_out/r05/replay.txt

####################################################################################################
# [25] r05_run25.sh
# round 5, GPU call 25: s_setprio around the strip kernel's unpairable instructions (csrc/issue_priority.py), with and without
# packed arithmetic: level solve A/B, then correctness of the candidates
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/issue_priority_ab.txt 2>&1
cat gpurun_out/r05/issue_priority_ab.txt

####################################################################################################
# [26] r05_run26.sh
# round 5, GPU call 26: the product library with the plain-instruction strip kernel + issue priority: GPU suite, bench lines
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_prio.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_prio.txt; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_prio.txt
python bench.py > gpurun_out/r05/bench_prio_default.json 2> gpurun_out/r05/bench_prio_default.err
python bench.py --workload cfg3_4096_grey --no-pmc --no-cpu-baseline --no-reference-baseline > gpurun_out/r05/bench_prio_grey.json 2>/dev/null
python bench.py --workload cfg2_1024_grey --no-pmc --no-cpu-baseline --no-reference-baseline --no-batch-leg > gpurun_out/r05/bench_prio_cfg2.json 2>/dev/null
python bench.py --workload cfg5_8192_grey --no-pmc --no-cpu-baseline --no-reference-baseline --no-batch-leg > gpurun_out/r05/bench_prio_cfg5.json 2>/dev/null
python - <<'PY'
import json
for n in ("default", "grey", "cfg2", "cfg5"):
    d = json.loads(open("gpurun_out/r05/bench_prio_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["config"]["workload"], d["value"], d.get("pairs_per_s"), d["ms_per_step"], "roofline", d["roofline"]["achieved"], d["roofline"].get("launch_ms"), "batch", (d.get("batch") or {}).get("pairs_per_s"), "ok", d["output_check"]["ok"])
PY

####################################################################################################
# [27] r05_run27.sh
# round 5, GPU call 27: the issue-priority filter (and no packed arithmetic) on EVERY kernel file, whole-pipeline rates + operators
set -e
mkdir -p gpurun_out/r05
WLS="cfg3_4096_gradient cfg2_1024_grey cfg4_1080p_batch cfg1_rub" bash tools/ab_bench.sh ab/base.so ab/allprio.so ab/allprio_nopk.so > gpurun_out/r05/all_priority_bench_ab.txt 2>&1
cat gpurun_out/r05/all_priority_bench_ab.txt
TOOL=tools/time_ops.py bash tools/ab_time.sh 4096 4096 > gpurun_out/r05/all_priority_ops_ab.txt 2>&1 || true
tail -n 60 gpurun_out/r05/all_priority_ops_ab.txt

####################################################################################################
# [28] r05_run28.sh
# round 5, GPU call 28: lanes in flight against the pipelined rate with the new strip kernel
set -e
mkdir -p gpurun_out/r05
for lanes in 2 3 4 6 8; do
  for wl in cfg3_4096_gradient cfg2_1024_grey; do
    python3 bench.py --workload $wl --pipeline $lanes --max-lanes $lanes --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl lanes $lanes pairs/s %.1f ms/step %.3f' % (d['pairs_per_s'], d['ms_per_step']))"
  done
done > gpurun_out/r05/lanes_sweep.txt 2>&1
cat gpurun_out/r05/lanes_sweep.txt

####################################################################################################
# [29] r05_run29.sh
# round 5, GPU call 29: three input rows in flight against two (strip kernel), level solve
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/rows_in_flight_ab.txt 2>&1
cat gpurun_out/r05/rows_in_flight_ab.txt

####################################################################################################
# [30] r05_run30.sh
# round 5, GPU call 30: the round's measurement pass again, with the plain-instruction strip kernel (profiles/r05_*)
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh bench trace trace_default > gpurun_out/r05/measure_final.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_final.txt; exit 1; }
cat gpurun_out/r05/measure_final.txt

####################################################################################################
# [31] r05_run31.sh
# round 5, GPU call 31: which build of the strip kernel for which launch -- the rule (lone up to one workgroup per CU), always
# paired, always lone: whole-pipeline rates and the lone pair's latency
set -e
mkdir -p gpurun_out/r05
for rep in 1 2; do
for kind in auto paired lone; do
  if [ $kind = auto ]; then unset FLOW2D_FUSED_KIND; else export FLOW2D_FUSED_KIND=$kind; fi
  echo "== strip kernel build: $kind"
  WLS="cfg3_4096_gradient cfg2_1024_grey cfg4_1080p_batch cfg1_rub" bash tools/ab_bench.sh ab/dual.so | awk 'NR<=4'
done
done > gpurun_out/r05/strip_kernel_kind_ab.txt 2>&1
cat gpurun_out/r05/strip_kernel_kind_ab.txt

####################################################################################################
# [32] r05_run32.sh
# round 5, GPU call 32: the final library (issue-priority strip kernel; log-derivative instances packed): GPU suite, fuzzers,
# the SOR workload's line, the default command as the driver runs it (wall time)
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_final.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_final.txt; exit 1; }
tail -n 2 gpurun_out/r05/gpu_tests_final.txt
timeout -k 10 900 python tools/fuzz_parity.py 2500 501 0 0.3 > gpurun_out/r05/fuzz_auto_issue_priority.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_issue_priority.txt
timeout -k 10 900 python tools/fuzz_parity.py 1500 502 2 0.35 > gpurun_out/r05/fuzz_strips_issue_priority.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_strips_issue_priority.txt
timeout -k 10 600 python tools/fuzz_reference.py 800 503 > gpurun_out/r05/fuzz_reference_issue_priority.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_issue_priority.txt
python bench.py --workload cfg3_4096_sor > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05/sor_bench.err
S=$(date +%s.%N); python bench.py > gpurun_out/r05_driver_style_bench_line.json 2> gpurun_out/r05/driver_style.err; E=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("gpurun_out/r05_cfg3_4096_sor_bench_line.json")); print("sor", d["pairs_per_s"], d["pairs_per_s_single"], json.dumps(d.get("sor_time_to_residual"))[:1200])
d=json.load(open("gpurun_out/r05_driver_style_bench_line.json")); print("default", d["pairs_per_s"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], "batch", d["batch"]["pairs_per_s"], "wall", $E-$S)
PY

####################################################################################################
# [33] r05_run33.sh
# round 5, GPU call 33: a register budget for the strip kernel below what two waves need (room for other lanes' kernels)
set -e
mkdir -p gpurun_out/r05
WLS="cfg3_4096_gradient cfg3_4096_grey cfg4_1080p_batch" bash tools/ab_bench.sh ab/dev.so ab/v224.so ab/v208.so > gpurun_out/r05/vgpr_budget_ab.txt 2>&1
cat gpurun_out/r05/vgpr_budget_ab.txt

####################################################################################################
# [34] r05_run34.sh
# round 5, GPU call 34: the planner's border / interior cost ratio with the new strip kernel
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/edge_cost_ab.txt 2>&1
cat gpurun_out/r05/edge_cost_ab.txt

####################################################################################################
# [35] r05_run35.sh
# round 5, GPU call 35: a long fuzz campaign on the final library
set -e
mkdir -p gpurun_out/r05
timeout -k 10 1000 python tools/fuzz_parity.py 6000 601 0 0.3 > gpurun_out/r05/fuzz_auto_long.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_long.txt

####################################################################################################
# [36] r05_run36.sh
# round 5, GPU call 36: strip kernel without the zero-increment branch and the interior row clamp: level solve A/B, quick parity
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/scalar_trim_ab.txt 2>&1
cat gpurun_out/r05/scalar_trim_ab.txt

####################################################################################################
# [37] r05_run37.sh
# round 5, GPU call 37: strip kernel with the exact start-up (no row test in the steady state), tile rule without border terms in
# interior strips: level solve A/B against call 36's builds, then the fused and kernel tests on the product build
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/scalar_trim2_ab.txt 2>&1
cat gpurun_out/r05/scalar_trim2_ab.txt
python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernels.py tests/test_gpu_reference.py -x -q 2>&1 | tail -n 3

####################################################################################################
# [38] r05_run38.sh
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

####################################################################################################
# [39] r05_run39.sh
# round 5, GPU call 39: kernel traces of the final library (profiles/r05_*_by_grid.txt), the SOR and driver-style lines
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh trace trace_default > gpurun_out/r05/measure_traces_final.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_traces_final.txt; exit 1; }
grep "fused_outer_kernel<5, [01], true, false, false>  *131072\|^kernel" gpurun_out/r05/measure_traces_final.txt | head
python bench.py --workload cfg3_4096_sor > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05/sor_bench.err
S=$(date +%s.%N); python bench.py > gpurun_out/r05_driver_style_bench_line.json 2> gpurun_out/r05/driver_style.err; E=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("gpurun_out/r05_cfg3_4096_sor_bench_line.json")); print("sor", d["pairs_per_s"], d["pairs_per_s_single"], d["pairs_per_s_incl_h2d"], d["value"], d["roofline"]["avg_launch_ms"])
d=json.load(open("gpurun_out/r05_driver_style_bench_line.json")); print("default", d["pairs_per_s"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["valu_issue_frac"], "batch", d["batch"]["pairs_per_s"], "wall", $E-$S)
PY

####################################################################################################
# [40] r05_run40.sh
# round 5, GPU call 40: where AUTO hands a level from the LDS tiles to the strips, with the faster strip kernel (pipelined rates)
set -e
mkdir -p gpurun_out/r05
for rep in 1 2; do
for px in 360000 250000 160000 90000 40000; do
  export FLOW2D_TILED_MAX_PIXELS=$px
  echo "== tiles up to $px pixels"
  WLS="cfg3_4096_gradient cfg1_rub cfg2_1024_grey" bash tools/ab_bench.sh ab/devfull.so 2>/dev/null | awk 'NR<=3'
done
done > gpurun_out/r05/tiled_max_pixels_ab.txt 2>&1
cat gpurun_out/r05/tiled_max_pixels_ab.txt

####################################################################################################
# [41] r05_run41.sh
# round 5, GPU call 41: how far apart two non-plain instructions may be to share one raised-priority run (issue_priority.py GAP)
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/priority_gap_ab.txt 2>&1
cat gpurun_out/r05/priority_gap_ab.txt

####################################################################################################
# [42] r05_run42.sh
# round 5, GPU call 42: the streaming medians with 32-bit plane offsets (and rows without mirror / clamp in interior strips) against
# 64-bit addresses; operator tests on the product build
set -e
mkdir -p gpurun_out/r05
TOOL=tools/time_ops.py bash tools/ab_time.sh 4096 > gpurun_out/r05/median_offsets_ab.txt 2>&1
grep "==\|median" gpurun_out/r05/median_offsets_ab.txt
python -m pytest tests/test_gpu_operators.py tests/test_gpu_reference.py -x -q 2>&1 | tail -n 2

####################################################################################################
# [43] r05_run43.sh
# round 5, GPU call 43: lanes for the launch-bound configuration (rub1 / rub2) and the batch configuration
set -e
mkdir -p gpurun_out/r05
for lanes in 4 6 8 12; do
  for wl in cfg1_rub cfg4_1080p_batch; do
    python3 bench.py --workload $wl --max-lanes $lanes --pipeline $lanes --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl lanes $lanes (used %s) pairs/s %.1f ms/step %.3f' % (d['config'].get('streams_per_gpu'), d['pairs_per_s'], d['ms_per_step']))"
  done
done > gpurun_out/r05/lanes_sweep_cfg1.txt 2>&1
cat gpurun_out/r05/lanes_sweep_cfg1.txt

####################################################################################################
# [44] r05_run44.sh
# round 5, GPU call 44: steps per lock-step group
set -e
mkdir -p gpurun_out/r05
run() { python3 bench.py --workload $1 --step-group $2 --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 step-group $2 pairs/s %.1f ms/step %.3f single %.2f ms  used GiB %s' % (d['pairs_per_s'], d['ms_per_step'], 1000/d['pairs_per_s_single'], d['device_memory']['used_gib']))" || echo "$1 step-group $2 failed"; }
{
for g in 4 8; do run cfg3_4096_gradient $g; done
for g in 2 4 8; do run cfg3_4096_grey $g; done
for g in 1 2; do run cfg5_8192_grey $g; done
for g in 2 4; do run cfg3_4096_sor $g; done
} > gpurun_out/r05/step_group_sweep3.txt 2>&1
grep step-group gpurun_out/r05/step_group_sweep3.txt

####################################################################################################
# [45] r05_run45.sh
# round 5, GPU call 45: the driver's command (--steps 20 --warmup 5) by steps per lock-step group
set -e
mkdir -p gpurun_out/r05
run() { python3 bench.py --gpus 1 --steps $3 --warmup 5 --workload $1 --step-group $2 --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 steps $3 step-group $2 pairs/s %.1f ms/step %.4f (%.4f-%.4f)' % (d['pairs_per_s'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max']))" || echo "$1 step-group $2 failed"; }
{
for k in 20 40 100; do for g in 1 2 4 8; do run cfg3_4096_gradient $g $k; done; done
} > gpurun_out/r05/step_group_by_steps.txt 2>&1
grep step-group gpurun_out/r05/step_group_by_steps.txt

####################################################################################################
# [46] r05_run46.sh
# round 5, GPU call 46: the driver's command with the new lock-step group sizes: all legs, wall time; the other workloads' lines
set -e
mkdir -p gpurun_out/r05
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/driver_cmd_groups.json 2> gpurun_out/r05/driver_cmd_groups.err; E=$(date +%s.%N)
python3 - <<PY
import json
d=json.load(open("gpurun_out/r05/driver_cmd_groups.json"))
print("driver cmd", d["pairs_per_s"], d["value"], d["ms_per_step"], d["config"]["steps_per_lock_step_group"], "single", d["pairs_per_s_single"], "h2d", d["pairs_per_s_incl_h2d"], "batch", d["batch"]["pairs_per_s"], "check", d["output_check"]["ok"], d["output_check"].get("oracle"), "mem", d["device_memory"]["used_gib"], "wall", $E-$S)
print(json.dumps(d.get("host_entry"))[:600])
PY
for wl in cfg2_1024_grey cfg1_rub cfg5_8192_grey; do
python3 bench.py --workload $wl --no-pmc --no-batch-leg --no-reference-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', d['pairs_per_s'], d['ms_per_step'], 'group', d['config']['steps_per_lock_step_group'], 'single', d['pairs_per_s_single'], 'h2d', d['pairs_per_s_incl_h2d'], 'check', d['output_check']['ok'], 'mem', d['device_memory']['used_gib'])"
done

####################################################################################################
# [47] r05_run47.sh
# round 5, GPU call 47: the batch leg before the main job against behind it (driver's command), and the main value without it
set -e
mkdir -p gpurun_out/r05
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1: main pairs/s %.1f  batch %s  h2d %.1f  launch_ms %s' % (d['pairs_per_s'], (d.get('batch') or {}).get('pairs_per_s'), d['pairs_per_s_incl_h2d'], d['roofline']['avg_launch_ms']))"; }
{
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline 2>/dev/null | show "batch leg first"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline --batch-leg-last 2>/dev/null | show "batch leg last "
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | show "no batch leg   "
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline 2>/dev/null | show "batch leg first"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline --batch-leg-last 2>/dev/null | show "batch leg last "
python3 bench.py --workload cfg4_1080p_batch --steps 64 --no-pmc --no-cpu-baseline --no-reference-baseline --no-host-entry-leg 2>/dev/null | show "cfg4 on its own"
} > gpurun_out/r05/batch_leg_first_ab.txt 2>&1
cat gpurun_out/r05/batch_leg_first_ab.txt

####################################################################################################
# [48] r05_run48.sh
# round 5, GPU call 48: bench lines of every workload with the new lock-step group sizes and the batch leg first; the driver's command
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh bench > gpurun_out/r05/measure_final3.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_final3.txt; exit 1; }
cat gpurun_out/r05/measure_final3.txt
python bench.py --workload cfg3_4096_sor > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05/sor_bench.err
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_driver_style_bench_line.json 2> gpurun_out/r05/driver_style.err; E=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("gpurun_out/r05_cfg3_4096_sor_bench_line.json")); print("sor", d["pairs_per_s"], d["pairs_per_s_single"], d["pairs_per_s_incl_h2d"], d["value"], d["roofline"]["avg_launch_ms"])
d=json.load(open("gpurun_out/r05_driver_style_bench_line.json")); print("driver cmd", d["pairs_per_s"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["valu_issue_frac"], "batch", d["batch"]["pairs_per_s"], "h2d", d["pairs_per_s_incl_h2d"], "ok", d["output_check"]["ok"], "wall", $E-$S)
PY

####################################################################################################
# [49] r05_run49.sh
# round 5, GPU call 49: config 4's line again (its host-entry leg had been given a batch object of single pairs)
set -e
mkdir -p gpurun_out/r05
python3 bench.py --workload cfg4_1080p_batch --no-batch-leg > gpurun_out/r05_cfg4_1080p_batch_bench_line.json 2> gpurun_out/r05/cfg4.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05_cfg4_1080p_batch_bench_line.json'))
print('cfg4', d['pairs_per_s'], d['ms_per_step_min'], d['ms_per_step_max'], 'single', d['pairs_per_s_single'], 'h2d', d['pairs_per_s_incl_h2d'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['effective_frac'], d['roofline']['valu_issue_frac'], d['output_check']['ok'])"

####################################################################################################
# [50] r05_run50.sh
# round 5, GPU call 50: GPU suite on the final tree
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_final3.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_final3.txt; exit 1; }
tail -n 2 gpurun_out/r05/gpu_tests_final3.txt

####################################################################################################
# [51] r05_run51.sh
# round 5, GPU call 51: long fuzz campaigns on the final tree (strips forced; against the reference's kernels), run_batch8 with one rank
set -e
mkdir -p gpurun_out/r05
timeout -k 10 540 python tools/fuzz_parity.py 3500 801 2 0.35 > gpurun_out/r05/fuzz_strips_long.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_strips_long.txt
timeout -k 10 420 python tools/fuzz_reference.py 1500 802 > gpurun_out/r05/fuzz_reference_long.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_long.txt
bash tools/run_batch8.sh --world 1 > gpurun_out/r05/run_batch8_world1.txt 2>&1; tail -n 2 gpurun_out/r05/run_batch8_world1.txt

####################################################################################################
# [52] r05_run52.sh
# round 5, GPU call 52: a lock-step group's finest level as ONE launch (strips as long as the group allows) against instance by instance
set -e
mkdir -p gpurun_out/r05
for g in 8 4 2; do
  echo "== steps per group $g"
  for so in ab/split.so ab/nosplit.so ab/split.so ab/nosplit.so; do
    FLOW2D_HIP_LIB=$PWD/$so python3 bench.py --gpus 1 --steps 40 --warmup 5 --step-group $g --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$so group $g pairs/s %.1f ms/step %.4f check %s' % (d['pairs_per_s'], d['ms_per_step'], d['output_check']['ok']))"
  done
done > gpurun_out/r05/finest_level_split_ab.txt 2>&1
cat gpurun_out/r05/finest_level_split_ab.txt

####################################################################################################
# [53] r05_run53.sh
# round 5, GPU call 53: SQ counters of the strip kernel after the issue-priority build (where the wave cycles go)
set -e
mkdir -p gpurun_out/r05
bash tools/pmc_sq.sh 4096 > gpurun_out/r05/fused_sq_counters.txt 2>&1
cat gpurun_out/r05/fused_sq_counters.txt

####################################################################################################
# [54] r05_run54.sh
# round 5, GPU call 54: what the strip waves wait for (s_waitcnt is 16 % of their cycles now): non-temporal stores / loads, three rows in flight
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/memory_wait_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/memory_wait_ab.txt | awk '/==/{n=$2} /constancy/{print n, $2, $7}'

####################################################################################################
# [55] r05_run55.sh
# round 5, GPU call 55: the strip kernel with every row folded onto cache-resident rows (timing probe): what memory costs it now
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/compute_only_probe.txt 2>&1
grep "==\|constancy" gpurun_out/r05/compute_only_probe.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7, "ms"}'

####################################################################################################
# [56] r05_run56.sh
# round 5, GPU call 56: hand-written loads / stores / waits in the strip kernel (2 and 3 row sets in flight): timing, then parity
set -e
mkdir -p gpurun_out/r05
bash tools/ab_time.sh 4096 4096 2 5 > gpurun_out/r05/manual_wait_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/manual_wait_ab.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7, "ms"}'
for v in manual2 manual3; do
  echo "== tests with ab/$v.so"
  FLOW2D_HIP_LIB=$PWD/ab/$v.so timeout -k 10 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -n 2
done

####################################################################################################
# [57] r05_run57.sh
# round 5, GPU call 57: do the waves' memory bursts collide?  starting them in four phases (level solve)
# rows in flight, against the global stores behind the lane mask and against the committed kernel; three rounds on one box
set -e
mkdir -p gpurun_out/r05
for i in 1 2; do bash tools/ab_time.sh 4096 4096 2 5; done > gpurun_out/r05/stagger_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/stagger_ab.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7, "ms"}' | sort | awk '{k=$1" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k]}' | sort

####################################################################################################
# [58] r05_run58.sh
# round 5, GPU call 57: do the waves' memory bursts collide?  starting them in four phases (level solve)
# rows in flight, against the global stores behind the lane mask and against the committed kernel; three rounds on one box
set -e
mkdir -p gpurun_out/r05
for i in 1 2; do bash tools/ab_time.sh 4096 4096 2 5; done > gpurun_out/r05/stagger_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/stagger_ab.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7, "ms"}' | sort | awk '{k=$1" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k]}' | sort

####################################################################################################
# [59] r05_run59.sh
# round 5, GPU call 59: more fuzzing of the final library (AUTO, groups and SOR included; against the reference's kernels)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 700 python tools/fuzz_parity.py 5000 901 0 0.3 > gpurun_out/r05/fuzz_auto_long2.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_long2.txt
timeout -k 10 400 python tools/fuzz_reference.py 1500 902 > gpurun_out/r05/fuzz_reference_long2.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_reference_long2.txt

####################################################################################################
# [60] r05_run60.sh
# round 5, GPU call 60: the planner's border / interior cost ratio once more, after the scalar trims (three rounds on one box)
set -e
mkdir -p gpurun_out/r05
for i in 1 2 3; do bash tools/ab_time.sh 4096 4096 2 5; done > gpurun_out/r05/edge_cost_ab2.txt 2>&1
grep "==\|constancy" gpurun_out/r05/edge_cost_ab2.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7}' | sort | awk '{k=$1" "$2" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k], "ms"}' | sort

####################################################################################################
# [61] r05_run61.sh
# round 5, GPU call 61: kernel traces of the final tree (the default command as the driver runs it; the eager single-stream traces)
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh trace trace_default > gpurun_out/r05/measure_traces_final2.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_traces_final2.txt; exit 1; }
grep "fused_outer_kernel<5, [01], true, false, false>  *131072" gpurun_out/r05/measure_traces_final2.txt | head

####################################################################################################
# [62] r05_run62.sh
# round 5, GPU call 62: which kernels the small configurations spend their time in now (groups of 32, four lanes, graph replay)
set -e
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
for wl in cfg1_rub cfg2_1024_grey; do
  rm -rf gpurun_out/trace_g32_$wl
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_g32_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg --steps 256 --repeats 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r05/trace_g32_$wl.err)
  python3 tools/summarize_trace.py gpurun_out/trace_g32_$wl/*/*kernel_trace.csv 14 > gpurun_out/r05/${wl}_groups_of_32_by_grid.txt
  rm -rf gpurun_out/trace_g32_$wl
  cat gpurun_out/r05/${wl}_groups_of_32_by_grid.txt
done

####################################################################################################
# [63] r05_run63.sh
# round 5, GPU call 63: four strip bodies (border selects of the touched borders only) against two, by the planner's cost ratio
set -e
mkdir -p gpurun_out/r05
for size in "4096 4096" "1920 1080" "1024 1024"; do
  echo "#### $size"
  for i in 1 2; do bash tools/ab_time.sh $size 2 5; done | grep "==\|constancy" | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7}' | sort | awk '{k=$1" "$2" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k], "ms"}' | sort
done > gpurun_out/r05/edge_xy_split_ab.txt 2>&1
cat gpurun_out/r05/edge_xy_split_ab.txt

####################################################################################################
# [64] r05_run64.sh
# round 5, GPU call 64: the driver's 20 steps over four lanes: groups of 8 + 8 + 4, of 5 x 4, of 4 x 5
set -e
mkdir -p gpurun_out/r05
run() { python3 bench.py --gpus 1 --steps $3 --warmup 5 --workload $1 --step-group $2 --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 steps $3 step-group $2 pairs/s %.1f ms/step %.4f (%.4f-%.4f)' % (d['pairs_per_s'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max']))" || echo "$1 step-group $2 failed"; }
{
for rep in 1 2; do for g in 8 5 4 10; do run cfg3_4096_gradient $g 20; done; done
} > gpurun_out/r05/step_group_20_steps.txt 2>&1
grep step-group gpurun_out/r05/step_group_20_steps.txt

####################################################################################################
# [65] r05_run65.sh
# round 5, GPU call 65: last check of the tree as committed: GPU suite, smoke(), the driver's command
set -e
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_last.txt 2>&1 || { tail -n 30 gpurun_out/r05/gpu_tests_last.txt; exit 1; }
tail -n 2 gpurun_out/r05/gpu_tests_last.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/driver_cmd_last.json 2> gpurun_out/r05/driver_cmd_last.err; E=$(date +%s.%N)
python3 - <<PY
import json
d=json.load(open("gpurun_out/r05/driver_cmd_last.json"))
print("driver cmd", d["pairs_per_s"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], "batch", d["batch"]["pairs_per_s"], "h2d", d["pairs_per_s_incl_h2d"], "ok", d["output_check"]["ok"], "wall", $E-$S)
PY

####################################################################################################
# [66] r05_run66.sh
# round 5, GPU call 66: the multi-rank branches of bench.py once more with the final defaults (4 ranks sharing the one GPU over gloo)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 600 python bench.py --gpus 4 --rehearse-on-one-gpu --steps 20 --warmup 5 > gpurun_out/r05/rehearse4.json 2> gpurun_out/r05/rehearse4.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r05/rehearse4.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d.get("rehearsal"), d["pairs_per_s"], "check", d["output_check"]["ok"], d["output_check"].get("oracle"), "batch", d["batch"]["pairs_per_s"], d["batch"]["gather"], "h2d", d["pairs_per_s_incl_h2d"], "mem", d["device_memory"]["used_gib"])
PY

####################################################################################################
# [67] r05_run67.sh
# round 5, GPU call 67: the round's final measurement pass, everything on ONE box: bench lines, eager traces, the default command's
# trace, the SOR line, the driver's command
set -e
mkdir -p gpurun_out/r05
ROUND=r05 bash tools/measure.sh bench trace trace_default > gpurun_out/r05/measure_final_pass.txt 2>&1 || { tail -n 20 gpurun_out/r05/measure_final_pass.txt; exit 1; }
grep "^cfg\|fused_outer_kernel<5, [01], true, false, false>  *131072" gpurun_out/r05/measure_final_pass.txt
python bench.py --workload cfg3_4096_sor > gpurun_out/r05_cfg3_4096_sor_bench_line.json 2> gpurun_out/r05/sor_bench.err
S=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_driver_style_bench_line.json 2> gpurun_out/r05/driver_style.err; E=$(date +%s.%N)
python - <<PY
import json
d=json.load(open("gpurun_out/r05_cfg3_4096_sor_bench_line.json")); print("sor", d["pairs_per_s"], d["pairs_per_s_single"], d["pairs_per_s_incl_h2d"], d["value"], d["roofline"]["avg_launch_ms"])
d=json.load(open("gpurun_out/r05_driver_style_bench_line.json")); print("driver cmd", d["pairs_per_s"], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["valu_issue_frac"], "batch", d["batch"]["pairs_per_s"], "h2d", d["pairs_per_s_incl_h2d"], "ok", d["output_check"]["ok"], "wall", $E-$S)
PY

####################################################################################################
# [68] r05_run68.sh
# round 5, GPU call 68: one more fuzz campaign on the final library (new seeds)
set -e
mkdir -p gpurun_out/r05
timeout -k 10 800 python tools/fuzz_parity.py 6000 1001 0 0.3 > gpurun_out/r05/fuzz_auto_long3.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_auto_long3.txt
timeout -k 10 300 python tools/fuzz_parity.py 1500 1002 2 0.35 > gpurun_out/r05/fuzz_strips_long3.txt 2>&1; tail -n 1 gpurun_out/r05/fuzz_strips_long3.txt

####################################################################################################
# [69] r05_run69.sh
# round 5, GPU call 69: the freshly built tree (clean __graft_entry__.build()): strip-kernel tests, smoke
set -e
python -m pytest tests/test_gpu_fused.py tests/test_gpu_flow.py -x -q 2>&1 | tail -n 2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver cmd', d['pairs_per_s'], d['roofline']['avg_launch_ms'], d['output_check']['ok'], d['batch']['pairs_per_s'])"

####################################################################################################
# [70] r05_run70.sh
# round 5, GPU call 70: two / three / six input rows in flight as register sets named by the step's ring position (global stores, one
# block per step: the compiler's waits become vmcnt(7) / (13) / (31)) against the committed kernel; four rounds on one box
set -e
mkdir -p gpurun_out/r05
for i in 1 2; do bash tools/ab_time.sh 4096 4096 2 5; done > gpurun_out/r05/row_sets_global_store_ab.txt 2>&1
grep "==\|constancy" gpurun_out/r05/row_sets_global_store_ab.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7}' | sort | awk '{k=$1" "$2" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k], "ms"}' | sort

####################################################################################################
# [71] r05_run71.sh
# round 5, GPU call 71: the strip kernel's loads and stores alone (no arithmetic: timing probe) with two / three / six rows in flight
set -e
mkdir -p gpurun_out/r05
for i in 1 2; do bash tools/ab_time.sh 4096 4096 2 5; done > gpurun_out/r05/memory_only_row_sets.txt 2>&1
grep "==\|constancy" gpurun_out/r05/memory_only_row_sets.txt | awk '/==/{n=$2} /constancy/{print n, "constancy", $2, $7}' | sort | awk '{k=$1" "$2" "$3; a[k]=a[k]" "$4} END{for(k in a) print k, a[k], "ms"}' | sort
