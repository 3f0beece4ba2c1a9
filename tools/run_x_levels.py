"""Developer workload: flow2d_resample_x_levels for every level of a 0.5 pyramid, both frames, a few launches (for
rocprofv3 passes and timing).  usage: python tools/run_x_levels.py [size] [launches]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = F.Context(0)
rng = np.random.default_rng(0)
a, b = (ctx.plane(n, n, rng.normal(0, 1, (n, n)).astype(np.float32)) for _ in range(2))
pa, pb = ctx.plane(n, n), ctx.plane(n, n)
widths, w = [], n
while w // 2 >= 4:
    w //= 2
    widths.append(w)
if len(sys.argv) > 4:  # keep only the levels with lo <= out_w <= hi
    widths = [x for x in widths if int(sys.argv[3]) <= x <= int(sys.argv[4])]
columns, col = [], 0
for lw in widths:
    columns.append(col)
    col += (lw + 3) // 4 * 4
ts = []
for _ in range(reps):
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    ctx.resample_x_levels(a, pa, n, n, widths, columns, b, pb)
    ctx.record(e1)
    ts.append(ctx.elapsed_ms(e0, e1) * 1e3)
print("resample_x_levels %d^2, %d levels %s, two frames: us per launch %s" % (n, len(widths), widths, ["%.1f" % t for t in ts]))
ctx.close()
