#!/bin/bash
# Round 6's GPU calls, one function per call (`gpurun -- bash tools/r06_gpu_calls.sh callN`): the literal command lists, kept so that a
# figure in profiles/r06_experiments/ can be traced to the command that produced it.  Variant libraries are built HERE first
# (ab/<name>.so: make -C cuda-flow2d_amd/csrc BUILD=build_<name> LIB=$PWD/ab/<name>.so EXTRA="-DFLOW2D_DEV_BUILD -DFLOW2D_FUSED_DEV ...").
# Steps of a call are joined with && so that a failed or timed-out GPU step starts no further GPU step.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R" || exit 1
export TMPDIR=/tmp

lib() { echo "$R/ab/$1.so"; }

# level solve 10 x 5 at 4096^2 (Grey, Gradient) of each named variant, twice round the list
ab_level() {
    for rep in 1 2; do
        for v in "$@"; do
            echo "== $v"
            FLOW2D_HIP_LIB=$(lib $v) timeout -k 10 120 python3 tools/time_sweep.py 4096 4096 2 2>&1 | grep "level solve" || return 1
        done
    done
}

# kernel trace of the level solve of one variant: the strip kernel's own duration (pack / unpack launches of the packed probe apart)
trace_level() {
    local v=$1
    rm -rf "$OUT/trace_$v"
    (cd /tmp && FLOW2D_HIP_LIB=$(lib $v) timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$v" -- \
        python3 "$R/tools/time_sweep.py" 4096 4096 2 > "$OUT/trace_$v.log" 2>&1) || { tail -5 "$OUT/trace_$v.log"; return 1; }
    python3 tools/summarize_trace.py "$OUT"/trace_$v/*/*kernel_trace.csv 12 > "$OUT/trace_${v}_by_grid.txt"
    rm -rf "$OUT/trace_$v"
    head -8 "$OUT/trace_${v}_by_grid.txt"
}

call1() {  # the probes of VERDICT r05 items 1 and 2: stall histogram, compute-only / memory-only, packed planes, no halo lanes (+ exchange cost)
    timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py -x -q > "$OUT/call1_fused_tests.log" 2>&1 || { tail -20 "$OUT/call1_fused_tests.log"; return 1; }
    tail -1 "$OUT/call1_fused_tests.log"
    ab_level dev compute memory packed_memory nohalo nohalo_x0 nohalo_x10 > "$OUT/call1_ab_level.txt" 2>&1 || { tail "$OUT/call1_ab_level.txt"; return 1; }
    cat "$OUT/call1_ab_level.txt"
    for v in stamps stamps_compute stamps_memory stamps_packed; do
        FLOW2D_HIP_LIB=$(lib $v) timeout -k 10 200 python3 tools/fused_wave_stamps.py 4096x4096 > "$OUT/call1_$v.txt" 2>&1 || { tail "$OUT/call1_$v.txt"; return 1; }
        grep -E "^==|launch span|stalls of|clock held" "$OUT/call1_$v.txt"
    done
    trace_level dev && trace_level packed && trace_level nohalo || return 1
    (cd /tmp && rocprofv3 -L > "$OUT/call1_counters_available.txt" 2>&1; true)
    grep -c . "$OUT/call1_counters_available.txt"
}

# one rocprofv3 --pmc pass (counters in their own run, kernel trace only) over tools/pmc_workload.py; prints the strip kernel's rows
pmc_pass() {
    local name=$1; shift
    rm -rf "$OUT/pmc_$name"
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/pmc_$name" -- python3 "$R/tools/pmc_workload.py" 4096 \
        > "$OUT/pmc_$name.log" 2>&1) || { echo "pass $name ($*): failed"; tail -3 "$OUT/pmc_$name.log"; rm -rf "$OUT/pmc_$name"; return 0; }
    cp "$OUT"/pmc_$name/*/*counter_collection.csv "$OUT/pmc_$name.csv" 2>/dev/null
    rm -rf "$OUT/pmc_$name"
    python3 - "$OUT/pmc_$name.csv" <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "fused_outer" not in r["Kernel_Name"]: continue
    key = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d[key]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in sorted(d.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-40s steady launches (mean of the last %d) %.6g" % (c, len(v) - 1, sum(v[1:]) / max(1, len(v) - 1)))
PY
}

call2() {  # wave stamps without the stall timing (full / compute-only / memory-only / no halo lanes / packed), raw stamps kept; vector-memory counters
    for v in stampsl stampsl_compute stampsl_memory stampsl_nohalo stampsl_packed; do
        FLOW2D_STALLS=0 FLOW2D_STAMPS_OUT="$OUT/raw_$v" FLOW2D_HIP_LIB=$(lib $v) timeout -k 10 200 python3 tools/fused_wave_stamps.py 4096x4096 > "$OUT/call2_$v.txt" 2>&1 \
            || { tail "$OUT/call2_$v.txt"; return 1; }
        grep -E "^==|launch span|clock held|SIMD done" "$OUT/call2_$v.txt"
    done
    ab_level dev compute memory > "$OUT/call2_ab_level.txt" 2>&1 || { tail "$OUT/call2_ab_level.txt"; return 1; }
    cat "$OUT/call2_ab_level.txt"
    {
        pmc_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
        pmc_pass sq2 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_ANY
        pmc_pass ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
        pmc_pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
        pmc_pass tcp2 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
        pmc_pass tcp3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum
        pmc_pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
        pmc_pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
        pmc_pass tcc3 TCC_BUSY_avr TCC_CYCLE_sum TCC_EA0_WRREQ_LEVEL_sum TCC_REQ_sum
    } > "$OUT/call2_pmc.txt" 2>&1
    cat "$OUT/call2_pmc.txt"
}

call3() {  # is the "memory cost" a clock effect?  full / compute-only / memory-only strips under SUSTAINED load: clock held and cycles per wave
    for v in stampsl stampsl_compute stampsl_memory; do
        FLOW2D_STALLS=0 FLOW2D_STAMPS_SUSTAIN=6 FLOW2D_HIP_LIB=$(lib $v) timeout -k 10 200 python3 tools/fused_wave_stamps.py 4096x4096 > "$OUT/call3_$v.txt" 2>&1 \
            || { tail "$OUT/call3_$v.txt"; return 1; }
        grep -E "^==|per launch|^   [0-9]" "$OUT/call3_$v.txt"
    done
}

call4() {  # equal work per XCD (side blocks dealt over the runs): old order (dev, stampsl) against new (dev2, stampsl2)
    timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py -x -q > "$OUT/call4_fused_tests.log" 2>&1 || { tail -20 "$OUT/call4_fused_tests.log"; return 1; }
    tail -1 "$OUT/call4_fused_tests.log"
    ab_level dev dev2 > "$OUT/call4_ab_level.txt" 2>&1 || { tail "$OUT/call4_ab_level.txt"; return 1; }
    cat "$OUT/call4_ab_level.txt"
    for v in stampsl stampsl2; do
        FLOW2D_STALLS=0 FLOW2D_HIP_LIB=$(lib $v) timeout -k 10 200 python3 tools/fused_wave_stamps.py 4096x4096 > "$OUT/call4_$v.txt" 2>&1 || { tail "$OUT/call4_$v.txt"; return 1; }
        grep -E "^==|launch span .* =|    0: |SIMD done" "$OUT/call4_$v.txt"
    done
}

call5() {  # call4 again + the medians with three-input selection (product) against round 5's comparator programs (ab/median_old.so)
    call4 || return 1
    timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference.py -x -q -k "median" > "$OUT/call5_median_tests.log" 2>&1 || { tail -20 "$OUT/call5_median_tests.log"; return 1; }
    tail -1 "$OUT/call5_median_tests.log"
    for rep in 1 2; do
        for so in ab/median_old.so cuda-flow2d_amd/csrc/libflow2d_hip.so; do
            echo "== $so"
            FLOW2D_HIP_LIB="$R/$so" timeout -k 10 120 python3 tools/time_ops.py 4096 2>&1 | grep -i "median" || return 1
        done
    done > "$OUT/call5_median_ab.txt" 2>&1
    cat "$OUT/call5_median_ab.txt"
}

call6() {  # the forked frame pyramid + packed build for lone contexts: flow tests, then the lone pair's latency on every workload
    timeout -k 10 900 python3 -m pytest tests/test_gpu_flow.py tests/test_gpu_fused.py -x -q > "$OUT/call6_flow_tests.log" 2>&1 || { tail -30 "$OUT/call6_flow_tests.log"; return 1; }
    tail -1 "$OUT/call6_flow_tests.log"
    for wl in cfg3_4096_gradient cfg2_1024_grey cfg1_rub cfg4_1080p_batch; do
        timeout -k 10 300 python3 bench.py --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>"$OUT/call6_$wl.err" |
            python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s pairs/s %8.1f  ms/step %7.3f  launch_ms %s  lone pair ms %s  (one stream, pipeline kernels: %s)' % ('$wl', d['pairs_per_s'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d.get('single_pair_latency_ms'), d.get('single_pair_latency_single_stream_ms')))" || { tail -5 "$OUT/call6_$wl.err"; return 1; }
    done
}

call7() {  # timeline of a lone config-3 pair, forked (lone 1) and on one stream (lone 0): where does the fork lose what it should gain?
    for lone in 1 0; do
        rm -rf "$OUT/lone_trace_$lone"
        (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/lone_trace_$lone" -- python3 "$R/tools/lone_pair_trace.py" cfg3_4096_gradient $lone 4 \
            > "$OUT/call7_lone$lone.log" 2>&1) || { tail -5 "$OUT/call7_lone$lone.log"; return 1; }
        grep "pair:" "$OUT/call7_lone$lone.log"
        python3 tools/lone_pair_timeline.py "$OUT"/lone_trace_$lone/*/*kernel_trace.csv 4 400 > "$OUT/call7_timeline_lone$lone.txt"
        rm -rf "$OUT/lone_trace_$lone"
        head -3 "$OUT/call7_timeline_lone$lone.txt"
    done
}

call8() {  # lone pair latency again (forked / one stream taking turns), the median tests incl. window 7, and the new bench fields
    timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference.py -x -q -k "median" > "$OUT/call8_median_tests.log" 2>&1 || { tail -20 "$OUT/call8_median_tests.log"; return 1; }
    tail -1 "$OUT/call8_median_tests.log"
    FLOW2D_HIP_LIB="$R/ab/median_old.so" timeout -k 10 120 python3 tools/time_ops.py 4096 2>&1 | grep -i "median"
    timeout -k 10 120 python3 tools/time_ops.py 4096 2>&1 | grep -i "median"
    for wl in cfg3_4096_gradient cfg2_1024_grey cfg4_1080p_batch; do
        timeout -k 10 300 python3 bench.py --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg 2>"$OUT/call8_$wl.err" > "$OUT/call8_$wl.json" || { tail -5 "$OUT/call8_$wl.err"; return 1; }
        python3 -c "
import json
d=json.load(open('$OUT/call8_$wl.json')); r=d['roofline']
print('%-22s pairs/s %8.1f  launch_ms %s  lone pair ms %s  (one stream, pipeline kernels: %s)  clock %s %s  valu frac %s at clock %s  hbm floor %s  memory-only %s compute-only %s' % ('$wl', d['pairs_per_s'], r.get('avg_launch_ms'), d.get('single_pair_latency_ms'), str(d.get('single_pair_latency_pipeline_kernels_ms')), r.get('shader_clock_ghz'), r.get('shader_clock_ghz_per_xcd'), r.get('valu_issue_frac'), r.get('valu_issue_frac_at_clock'), r.get('hbm_floor_us'), r.get('memory_only_us'), r.get('compute_only_us')))"
    done
}

call9() {  # the whole GPU suite on the current tree, then call8's lines again (three-way lone latency, clock under the finest level's solves)
    timeout -k 10 1150 python3 -m pytest tests -q -m gpu -x > "$OUT/call9_gpu_tests.log" 2>&1 || { tail -30 "$OUT/call9_gpu_tests.log"; return 1; }
    tail -2 "$OUT/call9_gpu_tests.log"
    call8
}

call10() {  # the streaming per-sweep kernel at strips of 16 (product) / 32 / 64 rows: time and FETCH_SIZE per launch
    for rows in 16 32 64; do
        echo "== strips of $rows rows"
        export FLOW2D_SWEEP_ROWS=$rows FLOW2D_HIP_LIB=$(lib dev2)
        timeout -k 10 120 python3 tools/time_per_sweep.py 4096 4096 2>&1 | grep -E "grey|gradient" || return 1
        rm -rf "$OUT/pmc_sweep"
        (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_sweep" -- python3 "$R/tools/time_per_sweep.py" 4096 4096 > "$OUT/pmc_sweep.log" 2>&1) || { tail -3 "$OUT/pmc_sweep.log"; return 1; }
        python3 - "$OUT"/pmc_sweep/*/*counter_collection.csv <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "sweep_stream" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        d[r["Kernel_Name"].split("(")[0][-34:]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print("   %-34s FETCH_SIZE per launch %.1f KB x 2 (gfx950 counts 128-byte requests as 64) = %.1f MB; algorithmic reads 8 x 64 MiB = 536.9 MB" % (k, sum(v) / len(v), 2 * sum(v) / len(v) / 1e3))
PY
        rm -rf "$OUT/pmc_sweep"
    done
    unset FLOW2D_SWEEP_ROWS FLOW2D_HIP_LIB
}

call11() {  # per-sweep strips A/B, then the three lone-pair variants on contexts of their own
    call10 > "$OUT/call10_per_sweep_rows_ab.txt" 2>&1 || { tail "$OUT/call10_per_sweep_rows_ab.txt"; return 1; }
    cat "$OUT/call10_per_sweep_rows_ab.txt"
    call8 2>&1 | grep -v "median\|passed"
}

call12() {  # stacked levels + one y-pass launch: kernel and flow tests, then the lone-pair lines
    timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py tests/test_gpu_reference.py -x -q > "$OUT/call12_tests.log" 2>&1 || { tail -30 "$OUT/call12_tests.log"; return 1; }
    tail -1 "$OUT/call12_tests.log"
    call8 2>&1 | grep -v "median\|passed"
}

fuzz() {  # random pipelines against the oracle (AUTO, strips forced) and against the reference's own kernels; usage: fuzz <seed base> [cases auto] [cases strips] [cases reference]
    local seed=${1:-600}
    timeout -k 10 480 python3 tools/fuzz_parity.py ${2:-1500} $seed 0 0.3 > "$OUT/fuzz_auto_$seed.txt" 2>&1; tail -1 "$OUT/fuzz_auto_$seed.txt"
    grep -c MISMATCH "$OUT/fuzz_auto_$seed.txt" && return 1
    timeout -k 10 300 python3 tools/fuzz_parity.py ${3:-600} $((seed + 1)) 2 0.3 > "$OUT/fuzz_strips_$seed.txt" 2>&1; tail -1 "$OUT/fuzz_strips_$seed.txt"
    grep -c MISMATCH "$OUT/fuzz_strips_$seed.txt" && return 1
    timeout -k 10 300 python3 tools/fuzz_reference.py ${4:-400} $((seed + 2)) > "$OUT/fuzz_reference_$seed.txt" 2>&1; tail -1 "$OUT/fuzz_reference_$seed.txt"
    grep -c MISMATCH "$OUT/fuzz_reference_$seed.txt" && return 1
    return 0
}

call13() {  # per-sweep strips of 16 / 32 rows again, 4096^2 and 8192^2, twice round (the 8192^2 line of the measurement pass read slower with 32)
    /bin/true
    for rep in 1 2; do
        for rows in 16 32; do
            for n in 4096 8192; do
                echo "== $n^2, strips of $rows rows"
                FLOW2D_SWEEP_ROWS=$rows FLOW2D_HIP_LIB=$(lib dev2) timeout -k 10 120 python3 tools/time_per_sweep.py $n $n 2>&1 | grep -E "grey|gradient" || return 1
            done
        done
    done
}

call14() {  # the y pass of all levels with every load of a thread in flight: tests, then its time in the traces of configs 3, 5, 4, 2
    timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py tests/test_gpu_fused.py -x -q > "$OUT/call14_tests.log" 2>&1 || { tail -30 "$OUT/call14_tests.log"; return 1; }
    tail -1 "$OUT/call14_tests.log"
    WLS="cfg3_4096_gradient cfg5_8192_grey cfg4_1080p_batch cfg2_1024_grey" ROUND=r06b bash tools/measure.sh trace > "$OUT/call14_trace.log" 2>&1 || { tail "$OUT/call14_trace.log"; return 1; }
    grep -h "resample_y_levels\|fused_outer_kernel<5, 1, true, false, false>     131072" "$OUT"/r06b_*_by_grid.txt
}

call15() {  # the y pass of all levels level by level; blur prefetch depth and registration gather order A/B
    timeout -k 10 120 python3 tools/time_y_levels.py 4096 7 || return 1
    timeout -k 10 120 python3 tools/time_y_levels.py 8192 11 || return 1
    for rep in 1 2; do
        for v in ops_base ops_b6 ops_b8 ops_reg; do
            echo "== $v"
            FLOW2D_HIP_LIB=$(lib $v) timeout -k 10 120 python3 tools/time_ops.py 4096 2>&1 | grep -E "gaussian|registration" || return 1
        done
    done
}

call16() {  # y-levels kernel with a flat grid, one batch of scalar loads, no lone cells: tests and its time level by level
    timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "resample or registration or gauss or convolution" > "$OUT/call16_tests.log" 2>&1 || { tail -30 "$OUT/call16_tests.log"; return 1; }
    tail -1 "$OUT/call16_tests.log"
    timeout -k 10 120 python3 tools/time_y_levels.py 4096 7 || return 1
    timeout -k 10 120 python3 tools/time_y_levels.py 8192 11 || return 1
}

call17() {  # up-sampling kernel with all loads of a thread in flight: tests, then against the previous commit's library
    timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference.py -x -q -k "resample" > "$OUT/call17_tests.log" 2>&1 || { tail -30 "$OUT/call17_tests.log"; return 1; }
    tail -1 "$OUT/call17_tests.log"
    for rep in 1 2; do
        for so in ab/ops_base.so cuda-flow2d_amd/csrc/libflow2d_hip.so; do
            echo "== $so"
            FLOW2D_HIP_LIB="$R/$so" timeout -k 10 120 python3 tools/time_ops.py 4096 2>&1 | grep -E "x and y from" || return 1
            FLOW2D_HIP_LIB="$R/$so" timeout -k 10 120 python3 tools/time_ops.py 2048 2>&1 | grep -E "x and y from" || return 1
        done
    done
}

call18() {  # lanes of the host-entry leg (uploads and downloads inside the bracket): 4 (as the device-resident region) / 6 / 8
    for rep in 1 2; do
        for lanes in 4 6 8; do
            timeout -k 10 300 python3 bench.py --no-pmc --no-oracle-check --no-cpu-baseline --no-reference-baseline --no-batch-leg --no-probe-builds --host-entry-lanes $lanes 2>"$OUT/call18.err" |
                python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['host_entry']
print('host-entry lanes $lanes: incl. H<->D %.1f pairs/s (%.1f GB/s each way), device-resident %.1f' % (h['pairs_per_s'], h['pcie_gbs_each_way'], d['pairs_per_s']))" || { tail -5 "$OUT/call18.err"; return 1; }
        done
    done
}

# whole-pipeline rate and lone-pair latency of the tree exported from the previous commit (ab/prev_tree: git archive HEAD + the current HIP
# library, host library built there) against the working tree, twice round
ab_trees() {
    for rep in 1 2; do
        for wl in "$@"; do
            for tree in ab/prev_tree .; do
                (cd "$R/$tree" && timeout -k 10 300 python3 bench.py --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline \
                    --no-reference-baseline --no-batch-leg --no-probe-builds 2>/dev/null) |
                    python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s %-14s pairs/s %8.1f  ms/step %7.3f  single_pair_ms %s' % ('$wl', '$tree', d['pairs_per_s'], d['ms_per_step'], d.get('single_pair_latency_ms')))" || return 1
            done
        done
    done
}

call19() {  # the previous level's flow up-sampled and frame 1 warped by it in one launch: tests, the two kernels against the one, whole pipelines
    timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py tests/test_gpu_reference.py -x -q > "$OUT/call19_tests.log" 2>&1 || { tail -30 "$OUT/call19_tests.log"; return 1; }
    tail -1 "$OUT/call19_tests.log"
    for n in 4096 2048 512; do timeout -k 10 120 python3 tools/time_ops.py $n 2>&1 | grep -E "registration" || return 1; done
    ab_trees cfg2_1024_grey cfg3_4096_grey cfg4_1080p_batch
}

call20() {  # the memory-side kernels (blur, medians, resample, warp) and the LDS tiles without the SLP vectoriser (noslp) / without packed fp32 and through the priority filter (nopk)
    for rep in 1 2; do
        for so in cuda-flow2d_amd/csrc/libflow2d_hip.so ab/noslp.so ab/nopk.so; do
            echo "== $so"
            FLOW2D_HIP_LIB="$R/$so" timeout -k 10 120 python3 tools/time_ops.py 4096 2>&1 | grep -E "^median|gaussian|registration \(one|levels|median 5 of" || return 1
            for n in 512 256 64; do FLOW2D_HIP_LIB="$R/$so" timeout -k 10 120 python3 tools/time_sweep.py $n $n 2 2>&1 | grep "level solve" || return 1; done
        done
    done
    WLS="cfg3_4096_gradient cfg2_1024_grey cfg4_1080p_batch" bash tools/ab_bench.sh cuda-flow2d_amd/csrc/libflow2d_hip.so ab/noslp.so ab/nopk.so
}

fuzz_halving() {  # random pipelines on 0.5 pyramids over frames of multiples of 32 (every level exactly twice the next) against the oracle; usage: fuzz_halving <seed> [cases auto] [cases strips]
    local seed=${1:-800}
    timeout -k 10 560 python3 tools/fuzz_parity.py ${2:-400} $seed 0 0.3 1.0 > "$OUT/fuzz_halving_auto_$seed.txt" 2>&1; tail -1 "$OUT/fuzz_halving_auto_$seed.txt"
    grep -c MISMATCH "$OUT/fuzz_halving_auto_$seed.txt" && return 1
    timeout -k 10 400 python3 tools/fuzz_parity.py ${3:-200} $((seed + 1)) 2 0.3 1.0 > "$OUT/fuzz_halving_strips_$seed.txt" 2>&1; tail -1 "$OUT/fuzz_halving_strips_$seed.txt"
    grep -c MISMATCH "$OUT/fuzz_halving_strips_$seed.txt" && return 1
    return 0
}

"$@"
