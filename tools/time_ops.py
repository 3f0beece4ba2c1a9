"""Scratch timing of the pyramid operators at one size (developer tool, not the bench contract).
usage: python tools/time_ops.py [size] [sigma]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


def timed(ctx, fn, reps=5):
    best = 1e9
    for _ in range(reps):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        fn()
        ctx.record(e1)
        best = min(best, ctx.elapsed_ms(e0, e1))
    return best * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
    ctx = F.Context(0)
    rng = np.random.default_rng(0)
    a, b, c, d = (ctx.plane(n, n, rng.normal(0, 1, (n, n)).astype(np.float32)) for _ in range(4))
    out, tmp = ctx.plane(n, n), ctx.plane(n, n)
    taps, radius = F.gaussian_kernel(sigma)
    plane_mb = n * n * 4 / 1e6
    rows = [
        ("median 3", lambda: ctx.median(a, n, n, 3, out), 2),
        ("median 5", lambda: ctx.median(a, n, n, 5, out), 2),
        ("median 7", lambda: ctx.median(a, n, n, 7, out), 2),
        ("gaussian blur sigma %.2f (radius %d)" % (sigma, radius), lambda: ctx.gaussian_blur(out, a, n, n, taps, radius), 2),
        ("add", lambda: ctx.add(out, a, n, n), 3),
        ("registration", lambda: ctx.registration(a, b, c, d, n, n, 1.0, 1.0, out), 5),
        ("resample x to 1/2", lambda: ctx.resample_x(a, tmp, n // 2, n, n), 1.5),
        ("resample y to 1/2 (half width)", lambda: ctx.resample_y(tmp, out, n // 2, n // 2, n), 0.75),
        ("resample x to 1/16", lambda: ctx.resample_x(a, tmp, n // 16, n, n), 1 + 1 / 16),
        ("resample y to 1/128 at width n/128 (coarsest level)", lambda: ctx.resample_y(a, out, n // 128, n // 128, n), 1 / 128),
        ("resample y to 1/32 at width n/32", lambda: ctx.resample_y(a, out, n // 32, n // 32, n), 1 / 32),
        ("resample x to 0.9", lambda: ctx.resample_x(a, tmp, int(n * 0.9), n, n), 1.9),
        ("resample x and y from 1/2 (two planes, one launch)", lambda: ctx.resample_xy(a, out, n // 2, n // 2, n, n, b, tmp), 2.5),
        ("median 5 of a + b (two planes)", lambda: ctx.add_median(a, b, n, n, 5, out, c, d, tmp), 6),
    ]
    if hasattr(ctx, "upsample_registration"):  # (u, v) from 1/2 and the warp by them, one launch; c, d hold a flow of a few pixels
        e, g = ctx.plane(n, n), ctx.plane(n, n)
        rows.append(("resample x and y from 1/2, then registration (two launches)",
                     lambda: (ctx.resample_xy(c, out, n // 2, n // 2, n, n, d, tmp), ctx.registration(a, b, out, tmp, n, n, 1.0, 1.0, e)), 6.5))
        rows.append(("up-sampling + registration (one launch)",
                     lambda: ctx.upsample_registration(c, d, n // 2, n // 2, out, tmp, a, b, n, n, 1.0, 1.0, g), 4.5))
    # the x pass of every level of a 0.5 pyramid for both frames in one trip (flow2d_resample_x_levels), then the y passes of
    # the coarsest levels out of the packed plane (long chains of rows, a handful of outputs)
    widths, w = [], n
    while w // 2 >= 4:
        w //= 2
        widths.append(w)
    columns, col = [], 0
    for lw in widths:
        columns.append(col)
        col += (lw + 3) // 4 * 4
    pa, pb = ctx.plane(n, n), ctx.plane(n, n)
    rows.append(("resample x, %d levels (%d..%d), two frames" % (len(widths), widths[0], widths[-1]),
                 lambda: ctx.resample_x_levels(a, pa, n, n, widths, columns, b, pb), 2 + 2.0 * sum(widths) / n))
    for lw in widths[-3:]:
        rows.append(("resample y to %d at width %d" % (lw, lw), lambda lw=lw: ctx.resample_y(pa, out, lw, lw, n), lw / n))
    for name, fn, planes in rows:
        us = timed(ctx, fn)
        print("%-42s %8.1f us  %6.2f TB/s algorithmic" % (name, us, planes * plane_mb / us))
    ctx.close()


if __name__ == "__main__":
    main()
