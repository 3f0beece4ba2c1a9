"""Developer tool: random whole-pipeline configurations, product against THE REFERENCE'S OWN KERNELS (oracle/_ref,
driven by oracle/ref_driver.cpp with the reference's launch sequence), bit for bit.  Grey: any size; Gradient and
LogDerivatives: level sizes that are multiples of the reference's 16x8 block at every level (off that grid its kernels
read an unwritten shared-memory slot).   usage (GPU box): python tools/fuzz_reference.py [cases] [seed]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")
from oracle import oracle as O
from oracle import ref_kernels as RK


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                                                 np.ascontiguousarray(b, np.float32).view(np.uint32))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    if not RK.available():
        sys.exit("oracle/_ref holds no reference kernels")
    bad, t0 = 0, time.time()
    for n in range(cases):
        constancy = int(rng.choice([0, 0, 1, 3]))
        if constancy == 0:
            w, h = int(rng.integers(8, 500)), int(rng.integers(8, 360))
            levels, scale = int(rng.integers(1, 12)), float(np.float32(rng.uniform(0.3, 0.95)))
            alpha = float(np.float32(10.0 ** rng.uniform(0.0, 2.0)))
        else:
            levels, scale = int(rng.integers(1, 5)), 0.5
            w, h = 16 * int(rng.integers(1, 5)) << (levels - 1), 8 * int(rng.integers(1, 7)) << (levels - 1)
            alpha = float(np.float32(10.0 ** rng.uniform(0.0, 2.0))) if constancy == 1 else 0.0005
        p = (levels, scale, int(rng.integers(1, 5)), int(rng.integers(1, 8)), alpha, 0.001, 0.001,
             int(rng.choice([3, 5, 7])), float(rng.choice([0.0, 0.45, 1.0, 1.5])))
        f0, f1 = O.synthetic_pair(w, h, float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), seed=int(rng.integers(1 << 30)),
                                  noise=True)
        flow = F.OpticalFlow(w, h, constancy)
        try:
            u, v, _ = flow.compute_flow(f0, f1, flow.params(*p))
        finally:
            flow.close()
        with RK.RefKernels(w, h) as R:
            ru, rv, _, _ = R.compute_flow(f0, f1, *p, constancy={0: RK.GREY, 1: RK.GRADIENT, 3: RK.LOG_DERIVATIVES}[constancy])
        if not (bits_equal(u, ru) and bits_equal(v, rv)):
            bad += 1
            print("MISMATCH case %d: %s" % (n, (w, h, constancy, p)))
        if n % 10 == 9:
            print("%d cases, %d mismatches, %.0f s" % (n + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
