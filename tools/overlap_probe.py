"""Do two fused solver kernels from different streams share the GPU?  (developer tool; run under
rocprofv3 --kernel-trace and feed the trace to tools/overlap_report.py)"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


def setup(ctx, n, rng):
    planes = [ctx.plane(n, n, rng.normal(0, 1, (n, n)).astype(np.float32)) for _ in range(4)]
    scratch = [ctx.plane(n, n).fill_bytes(0) for _ in range(6)]
    return planes, scratch


def main():
    a_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    b_n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rng = np.random.default_rng(0)
    ca, cb = F.Context(0), F.Context(0)
    pa, sa = setup(ca, a_n, rng)
    pb, sb = setup(cb, b_n, rng)
    for rep in range(6):
        ca.solve_level(*pa, *sa[:2], sa[2], sa[3], sa[4], sa[5], a_n, a_n, 1.0, 1.0, 35.0, 0.001, 0.001, 10, 5, 1, 2)
        for _ in range(6):
            cb.solve_level(*pb, *sb[:2], sb[2], sb[3], sb[4], sb[5], b_n, b_n, 1.0, 1.0, 35.0, 0.001, 0.001, 10, 5, 1, 2)
    ca.synchronize()
    cb.synchronize()
    ca.close()
    cb.close()


if __name__ == "__main__":
    main()
