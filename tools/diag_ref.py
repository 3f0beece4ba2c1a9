"""developer diagnostic: product vs CPU oracle vs the reference's own kernels on one configuration"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")
from oracle import oracle as O
from oracle import ref_kernels as RK
cases = [((61, 149), (9, 0.6, 3, 2, 80.0, 0.001, 0.001, 3, 0.0), 0),
         ((800, 720), (3, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5), 0),
         ((384, 256), (3, 0.5, 2, 3, 20.0, 0.001, 0.001, 3, 0.45), 3)]
for (w, h), p, c in cases:
    f0, f1 = O.synthetic_pair(w, h, 1.5, -0.75, seed=1, noise=(w * h <= 256 * 128))
    flow = F.OpticalFlow(w, h, c)
    u, v, _ = flow.compute_flow(f0, f1, flow.params(*p)); flow.close()
    oc = {0: 0, 1: 1, 3: 3}[c]
    try:
        ou, ov, _ = O.compute_flow(f0, f1, *p, oc)
    except Exception as e:
        ou = ov = None; print("oracle failed", e)
    with RK.RefKernels(w, h) as R:
        ru, rv, _, _ = R.compute_flow(f0, f1, *p, constancy={0: 0, 1: 1, 3: 2}[c])
    def cmp(a, b, name):
        if a is None or b is None: return
        d = np.abs(a.astype(np.float64) - b); bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
        print("  %-18s differing %d of %d, max |d| %.3g, first at %s" % (name, len(bad), a.size, d.max(), bad[:3].tolist()))
    print("max |u| ref %.4g" % float(np.abs(ru).max())); print((w, h), p, c, "levels run", F.host_lib().flow2d_host_max_warp_level_static(w, h, p[1]))
    cmp(u, ou, "product vs oracle"); cmp(u, ru, "product vs ref"); cmp(ou, ru, "oracle vs ref")
