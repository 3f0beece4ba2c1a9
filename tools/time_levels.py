"""Developer tool: level solve (10 x 5) by level size, fused strips against LDS tiles -- the numbers behind AUTO's
threshold (solve_level.hip, tiled_max_pixels).  usage (GPU box): python tools/time_levels.py [sizes, e.g. 512,640x480,...] [grid spacing h, default 1.0; e.g. 1.1 for a spacing that is no power of two]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


def main():
    sizes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["256", "384", "512", "640", "704", "768", "896", "1024", "1200", "1920x1080"]
    spacing = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    ctx = F.Context(0)
    rng = np.random.default_rng(0)
    for size in sizes:
        w, h = (int(t) for t in size.split("x")) if "x" in size else (int(size), int(size))
        planes = [ctx.plane(w, h, rng.normal(0, 1, (h, w)).astype(np.float32)) for _ in range(4)]
        du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h).fill_bytes(0) for _ in range(6))
        line = "%5d x %-5d" % (w, h)
        for constancy in (0, 1):
            for algo in (F.SOLVER_TILED, F.SOLVER_FUSED):
                best = 1e9
                for rep in range(30):
                    e0, e1 = ctx.event(), ctx.event()
                    ctx.record(e0)
                    ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, spacing, spacing, 35.0, 0.001, 0.001, 10, 5, constancy, algo)
                    ctx.record(e1)
                    best = min(best, ctx.elapsed_ms(e0, e1))
                line += "  %s %s %.3f" % ("grey" if constancy == 0 else "grad", "tiles" if algo == F.SOLVER_TILED else "strips", best)
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
