#!/bin/bash
# One job of flow2d_batch over the GPUs of a node with one PROCESS per GPU (BASELINE.json configs[3]: 64 pairs of 1920 x 1080,
# 8 per GPU), turnkey: N fresh child processes `flow2d_batch --rank R --world N --id-file DIR/id --run-id NONCE`, started by
# this shell -- which never touches a GPU, so nothing is exec'd or re-launched after GPU initialisation -- in a directory
# of the job's own (id file, side-channel files; removed at the end).  Prints rank 0's JSON line; exit code = the ranks'
# (the same on every rank: cuda-flow2d_amd/host/batch_driver.h).  With --bench it then runs `python bench.py --gpus N`
# (torch.distributed / RCCL, the driver's scaling command) for the same N.
#
# usage: tools/run_batch8.sh [--world N] [--bench] [flow2d_batch job flags: --pairs K --repeat R --width W --height H ...]
#        defaults: --world 8 --pairs 8*N --repeat 16 (config 4)
R=$(cd "$(dirname "$0")/.." && pwd)
TOOL="$R/cuda-flow2d_amd/host/flow2d_batch"
WORLD=8
BENCH=0
ARGS=()
while [ $# -gt 0 ]; do
    case "$1" in
        --world) WORLD=$2; shift 2 ;;
        --bench) BENCH=1; shift ;;
        *) ARGS+=("$1"); shift ;;
    esac
done
case " ${ARGS[*]} " in *" --pairs "*) ;; *) ARGS+=(--pairs $((8 * WORLD))) ;; esac
case " ${ARGS[*]} " in *" --repeat "*) ;; *) ARGS+=(--repeat 16) ;; esac
[ -x "$TOOL" ] || { echo "run_batch8.sh: $TOOL missing: make -C cuda-flow2d_amd/host" >&2; exit 1; }
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}  # dmabuf IPC: RCCL across processes needs it on this image
DIR=$(mktemp -d /tmp/flow2d_batch.XXXXXX) || exit 1
NONCE="$$-$(date +%s%N)"
PIDS=()
for ((r = 1; r < WORLD; r++)); do
    "$TOOL" --rank "$r" --world "$WORLD" --id-file "$DIR/id" --run-id "$NONCE" "${ARGS[@]}" > "$DIR/rank$r.out" 2> "$DIR/rank$r.err" &
    PIDS+=($!)
done
"$TOOL" --rank 0 --world "$WORLD" --id-file "$DIR/id" --run-id "$NONCE" "${ARGS[@]}"
CODE=$?
for p in "${PIDS[@]}"; do
    wait "$p"
    c=$?
    [ "$c" -gt "$CODE" ] && CODE=$c
done
for ((r = 1; r < WORLD; r++)); do [ -s "$DIR/rank$r.err" ] && sed "s/^/[rank $r] /" "$DIR/rank$r.err" >&2; done
rm -rf "$DIR"
if [ "$BENCH" = 1 ] && [ "$CODE" = 0 ]; then
    (cd "$R" && python bench.py --gpus "$WORLD" --steps 20 --warmup 3)
    CODE=$?
fi
exit $CODE
