"""Workload for the rocprofv3 --pmc passes that price the solver kernels' HBM traffic (profiles/traffic.json).

Launches, at 4096^2 (working set 12 x 64 MiB, far beyond the 256 MiB Infinity Cache):
  * two calibration kernels with exactly known bytes: add_2d (16 B per lane: 2 planes read, 1 written) and
    a 1-tap Gaussian row pass (4 B per lane: 1 plane read, 1 written) -- MI355X_MICROARCH.md says FETCH_SIZE
    halves wide coalesced reads on gfx950 and that other widths must be calibrated on a known byte count;
  * the fused outer-iteration kernel and the per-sweep kernels, Grey and Gradient.
Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/pmc_workload.py   (and again with WRITE_SIZE,
"TCC_HIT_sum TCC_MISS_sum" and "SQ_INSTS_VALU SQ_WAVES"; tools/pmc_passes.sh runs the four passes)
"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


def main():
    w = h = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ctx = F.Context(0)
    rng = np.random.default_rng(0)
    planes = [ctx.plane(w, h, rng.normal(0, 1, (h, w)).astype(np.float32)) for _ in range(4)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h).fill_bytes(0) for _ in range(6))
    one_tap = np.array([1.0], np.float32)
    for _ in range(5):
        ctx.add(tdu, planes[0], w, h)
        ctx.convolution_rows(tdv, planes[1], w, h, one_tap, 0)
    for constancy in (0, 1):
        for algo in (2, 1):
            # 4 outer iterations: the fused path's first launch of a level (du = dv = 0, planes not read) + 3 steady ones
            ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 4, 5, constancy, algo)
    ctx.synchronize()
    ctx.close()


if __name__ == "__main__":
    main()
