mkdir -p gpurun_out/r05
B() { timeout -k 10 400 python bench.py --steps 20 --no-cpu-baseline --no-reference-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-60s main %.1f pairs/s  batch leg %.1f pairs/s' % ('$*', d['pairs_per_s'], d['batch']['pairs_per_s']))"; }
B --no-pmc
B --no-pmc --no-host-entry-leg
B --no-pmc --no-oracle-check
B --no-pmc --no-host-entry-leg --no-oracle-check
B
