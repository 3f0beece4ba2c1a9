#!/bin/bash
# A round's measurement pass (run on the GPU box: [ROUND=r03] bash tools/measure.sh [tests|bench|trace|trace_default ...]).
# Every DESIGN.md number gets a file: gpurun_out/<round>_<workload>_bench_line.json (un-profiled bench line, its PMC
# figures from the rocprofv3 --pmc child passes bench.py runs itself) and gpurun_out/<round>_<workload>_by_grid.txt
# (eager single-stream kernel trace of the same workload).
# Steps are joined so that a failed or timed-out GPU step starts no further GPU step.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
ROUND=${ROUND:-r04}
mkdir -p "$OUT"
cd "$R" || exit 1
WHAT=${@:-tests bench trace}
WLS=${WLS:-"cfg3_4096_gradient cfg3_4096_grey cfg2_1024_grey cfg4_1080p_batch cfg5_8192_grey cfg1_rub"}
set -e
for what in $WHAT; do
    case $what in
    tests)
        timeout -k 10 1100 python -m pytest tests -q -m gpu -x > "$OUT/gpu_tests.log" 2>&1 || { tail -30 "$OUT/gpu_tests.log"; exit 1; }
        tail -2 "$OUT/gpu_tests.log" ;;
    bench)
        for wl in $WLS; do
            extra="--no-batch-leg"; [ $wl = cfg3_4096_gradient ] && extra=""
            timeout -k 10 400 python3 bench.py --workload $wl $extra > "$OUT/${ROUND}_${wl}_bench_line.json" 2> "$OUT/${ROUND}_${wl}_bench.err" \
                || { tail -5 "$OUT/${ROUND}_${wl}_bench.err"; exit 1; }
            python3 -c "
import json; d=json.load(open('$OUT/${ROUND}_${wl}_bench_line.json'))
print('$wl', 'value', d['value'], 'pairs/s', d['pairs_per_s'], 'single', d['pairs_per_s_single'], 'incl_h2d', d['pairs_per_s_incl_h2d'], 'ms/step', d['ms_per_step'], 'launch_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'per_sweep', (d['roofline']['per_sweep'] or {}).get('avg_launch_ms'), (d['roofline']['per_sweep'] or {}).get('frac'), (d['roofline']['per_sweep'] or {}).get('effective_frac'), 'check', d['output_check']['ok'])"
        done ;;
    trace)
        export TMPDIR=/tmp
        for wl in $WLS; do
            rm -rf "$OUT/trace_$wl"
            (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$wl" -- \
                python3 "$R/bench.py" --workload $wl --no-pmc --no-oracle-check --no-host-entry-leg --no-cpu-baseline --no-reference-baseline --no-batch-leg --no-probe-builds --steps 6 --warmup 2 --pipeline 1 --step-group 1 --no-graph \
                > "$OUT/${ROUND}_${wl}_bench_line_under_rocprof.json" 2> "$OUT/trace_$wl.err") || { tail -5 "$OUT/trace_$wl.err"; exit 1; }
            python3 tools/summarize_trace.py "$(ls -S "$OUT"/trace_$wl/*/*kernel_trace.csv | head -1)" 40 > "$OUT/${ROUND}_${wl}_by_grid.txt"
            cp "$(ls -S "$OUT"/trace_$wl/*/*kernel_stats.csv | head -1)" "$OUT/${ROUND}_${wl}_kernel_stats.csv"
            rm -rf "$OUT/trace_$wl"
            head -4 "$OUT/${ROUND}_${wl}_by_grid.txt"
        done ;;
    trace_default)  # the default bench command itself under the profiler (four lanes, graph replay)
        export TMPDIR=/tmp
        rm -rf "$OUT/trace_default"
        (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- \
            python3 "$R/bench.py" --no-pmc --no-probe-builds > "$OUT/${ROUND}_default_command_bench_line_under_rocprof.json" 2> "$OUT/trace_default.err") || { tail -5 "$OUT/trace_default.err"; exit 1; }
        python3 tools/summarize_trace.py "$(ls -S "$OUT"/trace_default/*/*kernel_trace.csv | head -1)" 30 > "$OUT/${ROUND}_default_command_by_grid.txt"
        cp "$(ls -S "$OUT"/trace_default/*/*kernel_stats.csv | head -1)" "$OUT/${ROUND}_default_command_kernel_stats.csv"
        rm -rf "$OUT/trace_default"
        head -5 "$OUT/${ROUND}_default_command_by_grid.txt" ;;
    esac
done
