"""Developer tool: random whole-pipeline configurations, product (single pairs and lock-step groups) against the CPU
oracle, bit for bit.  usage (GPU box): python tools/fuzz_parity.py [cases] [seed] [solver algorithm: 0 auto, 2 fused strips
on every level, ...] [share of cases in the opt-in red-black SOR mode, default 0.2: omega drawn from (0.3, 1.95)]
[share of cases on a 0.5 pyramid over a frame of up to 1400 x 1000 rounded to a multiple of 32 -- every level exactly twice the next:
the doubling form of the one-launch warp, the power-of-two x passes; default 0, which leaves the random stream of earlier campaigns as it was]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")
from oracle import oracle as O


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    algorithm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    sor_share = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
    halving_share = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    ctx = F.Context(0)
    bad = 0
    t0 = time.time()
    for n in range(cases):
        w, h = int(rng.integers(8, 700)), int(rng.integers(8, 500))
        constancy = int(rng.choice([0, 0, 1, 2]))
        p = (int(rng.integers(1, 14)), float(np.float32(rng.uniform(0.3, 0.95))), int(rng.integers(1, 5)),
             int(rng.integers(1, 9)), float(np.float32(10.0 ** rng.uniform(-1.0, 2.0))), 0.001, 0.001,
             int(rng.choice([1, 3, 5, 7])), float(rng.choice([0.0, 0.45, 1.0, 1.5, 2.9])))
        G = int(rng.choice([1, 1, 2, 3, 5]))
        if halving_share > 0.0 and rng.uniform() < halving_share:
            w, h = 32 * int(rng.integers(1, 44)), 32 * int(rng.integers(1, 32))
            p = (p[0], 0.5) + p[2:]
        omega = float(np.float32(rng.uniform(0.3, 1.95))) if rng.uniform() < sor_share else 0.0
        pairs = [O.synthetic_pair(w, h, float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), seed=int(rng.integers(1 << 30)),
                                  noise=bool(rng.integers(2))) for _ in range(G)]
        batch = F.OpticalFlowBatch(w, h, constancy, lanes=1, group_size=G)
        try:
            planes = [ctx.plane(w, h * G, np.vstack([q[0] for q in pairs])), ctx.plane(w, h * G, np.vstack([q[1] for q in pairs])),
                      ctx.plane(w, h * G).fill_bytes(0x7f), ctx.plane(w, h * G).fill_bytes(0x7f)]
            batch.use_graph(bool(rng.integers(2)))
            try:
                batch.compute_flow_batch_device(*[[q.ptr] for q in planes], batch.params(*p, algorithm, sor_omega=omega))
            except F.Flow2DError as e:
                print("case %d refused (%s): %s" % (n, e, (w, h, constancy, p, G, omega)))
                continue
            batch.synchronize()
            u, v = planes[2].download(), planes[3].download()
            for k, (f0, f1) in enumerate(pairs):
                ou, ov, _ = O.compute_flow(f0, f1, *p, constancy, sor_omega=omega)
                if not (np.array_equal(u[k * h:(k + 1) * h], ou, equal_nan=True) and
                        np.array_equal(v[k * h:(k + 1) * h], ov, equal_nan=True)):
                    bad += 1
                    print("MISMATCH case %d pair %d: %s" % (n, k, (w, h, constancy, p, G, omega)))
            for q in planes:
                q.free()
        finally:
            batch.close()
        if n % 20 == 19:
            print("%d cases, %d mismatches, %.0f s" % (n + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
