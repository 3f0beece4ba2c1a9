"""Developer experiment: does the relative alignment of the ten planes of a per-sweep launch matter?  Every plane is
allocated with a few spare rows and used at base + i * skew bytes (i = plane number).  hipMalloc hands out large
allocations on 2 MiB boundaries, so with skew 0 the same (x, y) of all ten planes maps to the same HBM channel and bank.
usage: python tools/sweep_alias_ab.py [size] [skew bytes ...]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


class View:
    def __init__(self, plane, offset):
        self.ptr, self.pitch, self.width, self.height, self.ctx = plane.ptr + offset, plane.pitch, plane.width, plane.height, plane.ctx


n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
skews = [int(a) for a in sys.argv[2:]] or [0, 256, 1024, 4096, 4352, 65536 + 256, 1 << 20]
ctx = F.Context(0)
rng = np.random.default_rng(0)
spare = 80
base = [ctx.plane(n, n + spare, rng.normal(0, 1, (n + spare, n)).astype(np.float32)) for _ in range(4)] + \
       [ctx.plane(n, n + spare).fill_bytes(0) for _ in range(6)]
for skew in skews:
    assert skew * 9 <= spare * base[0].pitch and skew % 16 == 0
    f0, f1, u, v, du, dv, phi, ksi, tdu, tdv = (View(p, i * skew) for i, p in enumerate(base))
    ctx.compute_phi_ksi(f0, f1, u, v, du, dv, n, n, 1.0, 1.0, 0.001, 0.001, phi, ksi)
    line = "skew %8d B:" % skew
    for name, constancy in (("grey", 0), ("gradient", 1)):
        ms = []
        for _ in range(6):
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            for k in range(10):
                a, b = ((du, dv), (tdu, tdv)) if k % 2 == 0 else ((tdu, tdv), (du, dv))
                ctx.solve_sweep(f0, f1, u, v, a[0], a[1], phi, ksi, n, n, 1.0, 1.0, 35.0, b[0], b[1], constancy)
            ctx.record(e1)
            ms.append(ctx.elapsed_ms(e0, e1) / 10)
        us = float(np.mean(ms[2:])) * 1e3
        line += "  %s %6.1f us (%.2f TB/s)" % (name, us, 40.0 * n * n / us / 1e6)
    print(line, flush=True)
ctx.close()
