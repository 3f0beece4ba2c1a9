"""Developer tool: the per-sweep Jacobi kernels (flow2d_solve_2d / _grad) alone at one size: us per launch by HIP events
around ten back-to-back launches, and the algorithmic rate (40 B per pixel).  usage: python tools/time_per_sweep.py [w] [h]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")

w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else w
ctx = F.Context(0)
rng = np.random.default_rng(0)
f0, f1, u, v = (ctx.plane(w, h, rng.normal(0, 1, (h, w)).astype(np.float32)) for _ in range(4))
du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h).fill_bytes(0) for _ in range(6))
ctx.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, 1.0, 1.0, 0.001, 0.001, phi, ksi)
for name, constancy in (("grey", 0), ("gradient", 1), ("untiled", 2), ("log", 3)):
    ms = []
    for _ in range(6):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for k in range(10):
            a, b = ((du, dv), (tdu, tdv)) if k % 2 == 0 else ((tdu, tdv), (du, dv))
            ctx.solve_sweep(f0, f1, u, v, a[0], a[1], phi, ksi, w, h, 1.0, 1.0, 35.0, b[0], b[1], constancy)
        ctx.record(e1)
        ms.append(ctx.elapsed_ms(e0, e1) / 10)
    us = float(np.mean(ms[2:])) * 1e3
    print("%dx%d %-8s sweep %7.1f us per launch  %.2f TB/s algorithmic (40 B per pixel)" % (w, h, name, us, 40.0 * w * h / us / 1e6))
# opt-in red-black SOR: one iteration = two half-sweep launches, in place
for name, constancy in (("grey", 0), ("gradient", 1), ("untiled", 2)):
    ms = []
    for _ in range(6):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for k in range(5):
            ctx.sor_iteration(f0, f1, u, v, du, dv, phi, ksi, w, h, 1.0, 1.0, 35.0, 1.5, constancy)
        ctx.record(e1)
        ms.append(ctx.elapsed_ms(e0, e1) / 5)
    us = float(np.mean(ms[2:])) * 1e3
    print("%dx%d %-8s SOR iteration (two half-sweep launches) %7.1f us" % (w, h, name, us))
ctx.close()
