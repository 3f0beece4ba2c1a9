"""Host <-> device copy rates from page-locked Data2D images through the C-ABI (developer tool): upload alone, download
alone, both at once on two streams, 4096^2 planes.  usage: python tools/pcie_pinned.py"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


def main():
    w = h = 4096
    L = F.hip_lib()
    up, down = F.Context(0), F.Context(0)
    imgs = [F.HostImage(w, h, True) for _ in range(4)]
    planes = [up.plane(w, h) for _ in range(4)]
    nbytes = w * h * 4

    def h2d(c, n):
        for k in range(n):
            L.flow2d_copy_h2d_2d(c.handle, planes[k % 2].ptr, planes[0].pitch, imgs[k % 2].array.ctypes.data, w * 4, w * 4, h)

    def d2h(c, n):
        for k in range(n):
            L.flow2d_copy_d2h_2d(c.handle, imgs[2 + k % 2].array.ctypes.data, w * 4, planes[2 + k % 2].ptr, planes[0].pitch, w * 4, h)

    for name, fn in (("upload alone", lambda: h2d(up, 16)), ("download alone", lambda: d2h(down, 16)),
                     ("both at once", lambda: (h2d(up, 16), d2h(down, 16)))):
        fn()
        up.synchronize(), down.synchronize()
        t0 = time.perf_counter()
        fn()
        up.synchronize(), down.synchronize()
        t = time.perf_counter() - t0
        print("%-15s %.1f GB/s per direction (16 x 64 MiB in %.2f ms)" % (name, 16 * nbytes / t / 1e9, t * 1e3))
    for q in imgs:
        q.close()
    up.close(), down.close()


if __name__ == "__main__":
    main()
