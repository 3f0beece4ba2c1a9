"""Developer tool (round 6): the y pass of all pyramid levels in one launch (flow2d_resample_y_levels) at one frame size -- all levels,
each level alone, the fine and the coarse levels apart.  usage: python tools/time_y_levels.py [size] [levels]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    ctx = F.Context(0)
    rng = np.random.default_rng(0)
    pa, pb = (ctx.plane(n, n, rng.normal(0, 1, (n, n)).astype(np.float32)) for _ in range(2))
    oa, ob = ctx.plane(n, n), ctx.plane(n, n)
    sizes = [n >> l for l in range(levels, 0, -1)]  # coarsest first, like the pyramid
    cols, rows, c, r = [], [], 0, 0
    for s in sizes:
        cols.append(c), rows.append(r)
        c += (s + 3) // 4 * 4
        r += s

    def timed(idx, reps=8):
        best = 1e9
        for _ in range(reps):
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            ctx.resample_y_levels(pa, oa, n, [sizes[i] for i in idx], [sizes[i] for i in idx], [cols[i] for i in idx], [rows[i] for i in idx], pb, ob)
            ctx.record(e1)
            best = min(best, ctx.elapsed_ms(e0, e1))
        mb = sum(2 * 4 * (sizes[i] * n + sizes[i] * sizes[i]) for i in idx) / 1e6
        return best * 1e3, mb

    for name, idx in [("all %d levels" % levels, list(range(levels)))] + [("level to %d" % sizes[i], [i]) for i in range(levels)] + \
            [("the three finest", list(range(levels - 3, levels))), ("all but the three finest", list(range(levels - 3)))]:
        us, mb = timed(idx)
        print("%-28s %8.1f us  %7.1f MB  %.2f TB/s" % (name, us, mb, mb / us))
    ctx.close()


if __name__ == "__main__":
    main()
