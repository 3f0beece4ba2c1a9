"""Developer tool: distribution of wave start/end times of one fused outer-iteration launch
(FLOW2D_FUSED_STAMPS=1 must be set before the library loads).  usage: python tools/wave_stamps.py [W] [H] [constancy]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

os.environ["FLOW2D_FUSED_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    h = int(sys.argv[2]) if len(sys.argv) > 2 else w
    constancy = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    ctx = F.Context(0)
    rng = np.random.default_rng(0)
    planes = [ctx.plane(w, h, rng.normal(0, 1, (h, w)).astype(np.float32)) for _ in range(4)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h).fill_bytes(0) for _ in range(6))
    for _ in range(3):
        ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 4, 5, constancy, 2)
    ctx.synchronize()
    L = F.hip_lib()
    cap = 1 << 16
    buf = np.zeros((cap, 4), np.uint64)
    n = C.c_size_t()
    L.flow2d_debug_fused_stamps.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    rc = L.flow2d_debug_fused_stamps(buf.ctypes.data, cap, C.byref(n))
    assert rc == 0, rc
    s = buf[: n.value]
    index = np.arange(n.value)[s[:, 1] > 0]
    s = s[s[:, 1] > 0]
    t0 = s[:, 0].min()
    start = (s[:, 0] - t0).astype(np.float64) / 100.0  # us
    end = (s[:, 1] - t0).astype(np.float64) / 100.0
    life = end - start
    print("%d waves; kernel span %.1f us; start: median %.1f p99 %.1f max %.1f us" %
          (len(s), end.max(), np.median(start), np.percentile(start, 99), start.max()))
    print("lifetime us: min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f" %
          (life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max()))
    print("end us: p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f" %
          (np.percentile(end, 10), np.median(end), np.percentile(end, 90), np.percentile(end, 99), end.max()))
    hw = s[:, 2]
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    xcc = s[:, 3] & 15
    unit = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
    uniq, counts = np.unique(unit, return_counts=True)
    print("distinct SIMDs used %d; waves per SIMD histogram %s" % (len(uniq), dict(zip(*np.unique(counts, return_counts=True)))))
    # lifetime by number of co-resident waves on the SIMD
    cmap = dict(zip(uniq, counts))
    per = np.array([cmap[x] for x in unit])
    for k in sorted(set(per)):
        print("  waves on a SIMD hosting %d: %d waves, median lifetime %.1f us, median end %.1f" %
              (k, (per == k).sum(), np.median(life[per == k]), np.median(end[per == k])))
    # by position: strips on the image border run the variant with the reflect selects
    valid = 64 - 2 * 6
    strips_x = -(-w // valid)
    gx = -(-strips_x // 4)
    strip_x = (index // 4) % gx * 4 + index % 4
    strip_y = (index // 4) // gx
    ny = strip_y.max() + 1
    edge = (strip_x == 0) | (strip_x >= strips_x - 2) | (strip_y == 0) | (strip_y == ny - 1)
    print("grid %d x %d strips; edge waves %d: median lifetime %.1f; interior %d: median %.1f p90 %.1f max %.1f" %
          (strips_x, ny, edge.sum(), np.median(life[edge]), (~edge).sum(), np.median(life[~edge]),
           np.percentile(life[~edge], 90), life[~edge].max()))
    for k in sorted(set(xcc)):
        m = xcc == k
        print("  xcc %d: %4d waves, median lifetime %.1f, max end %.1f" % (k, m.sum(), np.median(life[m]), end[m].max()))
    # pairs on one SIMD: who is the partner?
    order = np.argsort(unit, kind="stable")
    firsts, lasts, mixed = [], [], 0
    i = 0
    while i < len(order):
        j = i
        while j + 1 < len(order) and unit[order[j + 1]] == unit[order[i]]:
            j += 1
        if j == i + 1:
            a, b = order[i], order[j]
            firsts.append(min(end[a], end[b]))
            lasts.append(max(end[a], end[b]))
            mixed += int(edge[a] != edge[b])
        i = j + 1
    firsts, lasts = np.array(firsts), np.array(lasts)
    print("pairs: %d (%d edge+interior); first-to-finish median %.1f, last-to-finish median %.1f p90 %.1f max %.1f" %
          (len(firsts), mixed, np.median(firsts), np.median(lasts), np.percentile(lasts, 90), lasts.max()))
    print("row of strips -> median lifetime:", " ".join("%d:%.0f" % (y, np.median(life[strip_y == y])) for y in range(ny)))
    np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "wave_stamps.npy"), s)
    ctx.close()


if __name__ == "__main__":
    main()
