#!/bin/bash
# round 5, GPU call 69: the freshly built tree (clean __graft_entry__.build()): strip-kernel tests, smoke
set -e
python -m pytest tests/test_gpu_fused.py tests/test_gpu_flow.py -x -q 2>&1 | tail -n 2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-reference-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver cmd', d['pairs_per_s'], d['roofline']['avg_launch_ms'], d['output_check']['ok'], d['batch']['pairs_per_s'])"
