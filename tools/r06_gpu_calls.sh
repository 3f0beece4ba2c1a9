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

"$@"
