"""Scratch timing of the solver kernels at one size (developer tool, not the bench contract)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    h = int(sys.argv[2]) if len(sys.argv) > 2 else w
    algos = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1]
    inner = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    ctx = F.Context(0)
    print(ctx.device_name(), ctx.mem_info())
    rng = np.random.default_rng(0)
    planes = [ctx.plane(w, h, rng.normal(0, 1, (h, w)).astype(np.float32)) for _ in range(4)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h).fill_bytes(0) for _ in range(6))
    for constancy in (0, 1):
        for algo in algos:
            best = 1e9
            for rep in range(12):  # the first calls run on a cold GPU: report the last and the best
                e0, e1 = ctx.event(), ctx.event()
                ctx.record(e0)
                ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 10, inner,
                                constancy, algo)
                ctx.record(e1)
                ms = ctx.elapsed_ms(e0, e1)
                best = min(best, ms)
            bytes_ = w * h * 10 * (32 + 40 * inner)
            print("constancy %d algo %d: level solve %.3f ms (best %.3f) -> %.1f Mpix-iters/s, %.2f TB/s algorithmic (fallback waves so far %d)" %
                  (constancy, algo, ms, best, w * h * 10 * inner / ms / 1e3, bytes_ / ms / 1e9, ctx.fused_fallbacks()))
    ctx.close()


if __name__ == "__main__":
    main()
