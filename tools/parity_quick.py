"""quick parity of the fused kernel (algorithm 2) against the oracle, grey + gradient, 10x5, at a few sizes"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
F = importlib.import_module("cuda-flow2d_amd")
from oracle import oracle as O
ctx = F.Context(0)
ok = True
for (w, h) in ((300, 200), (1024, 512)):
    rng = np.random.default_rng(1)
    f0, f1 = O.synthetic_pair(w, h, 1.5, -0.75, seed=1, noise=True)
    u = rng.normal(0, 1, (h, w)).astype(np.float32); v = rng.normal(0, 1, (h, w)).astype(np.float32)
    for constancy in (0, 1):
        planes = [ctx.plane(w, h, a) for a in (f0, f1, u, v)]
        du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h).fill_bytes(0) for _ in range(6))
        rdu, rdv = ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 4, 5, constancy, 2)
        odu, odv, _, _ = O.solve_level(f0, f1, u, v, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 4, 5, constancy)
        a, b = rdu.download(w, h), rdv.download(w, h)
        eq = np.array_equal(a, odu) and np.array_equal(b, odv)
        print(w, h, "constancy", constancy, "bit-equal" if eq else "DIFF max %g" % np.abs(a - odu).max())
        ok &= eq
sys.exit(0 if ok else 1)
