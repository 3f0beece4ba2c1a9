"""Regenerates the golden vectors of tests/golden/ from the CPU oracle (oracle/flow2d_oracle.c).

Provenance: the reference ships no golden outputs and cannot be built in this image, so these vectors
come from the oracle, whose rub1/rub2 statistics match the anchors SURVEY.md 8(c) recorded from the
reference's own sources (tests/test_oracle.py).  They pin the oracle against silent drift and give
the GPU tests fixed expected outputs.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()


def rub():
    d = os.path.join(os.path.dirname(HERE), "data")
    r1 = np.fromfile(os.path.join(d, "rub1.raw"), np.uint8).reshape(388, 584).astype(np.float32)
    r2 = np.fromfile(os.path.join(d, "rub2.raw"), np.uint8).reshape(388, 584).astype(np.float32)
    return r1, r2


def main():
    out = {}
    # config 1: rub1 <-> rub2, settings.xml solver values
    r1, r2 = rub()
    u, v, _ = O.compute_flow(r1, r2, 20, 0.9, 20, 5, 3.5, 0.001, 0.001, 5, 0.45)
    out["rub_settings_u_sub4"] = u[::4, ::4]
    out["rub_settings_v_sub4"] = v[::4, ::4]
    out["rub_settings_sha"] = np.array([sha(u), sha(v)])
    # a short rub run (8 levels, 3 x 5 sweeps): stride-2 sub-grid + sha256 of the full fields
    u, v, _ = O.compute_flow(r1, r2, 8, 0.8, 3, 5, 3.5, 0.001, 0.001, 5, 0.45)
    out["rub_short_u_sub2"] = u[::2, ::2]
    out["rub_short_v_sub2"] = v[::2, ::2]
    out["rub_short_sha"] = np.array([sha(u), sha(v)])

    # per-stage vectors on a size that is not a multiple of any tile, Grey and Gradient
    f0, f1 = O.synthetic_pair(100, 70, 1.5, -0.75)
    rng = np.random.default_rng(2024)
    f1 = (f1 + rng.uniform(-1, 1, f1.shape)).astype(np.float32)
    out["small_f0"], out["small_f1"] = f0, f1
    for name, constancy in (("grey", 0), ("grad", 1)):
        stages = {}

        def dump(tag, level, plane, stages=stages):
            stages["%s_L%d" % (tag, level)] = plane.copy()

        u, v, _ = O.compute_flow(f0, f1, 6, 0.8, 2, 3, 3.5, 0.001, 0.001, 5, 0.45, constancy, dump=dump)
        out["small_%s_u" % name], out["small_%s_v" % name] = u, v
        for k in ("blur0_L-1", "frame0_res_L3", "warped_L2", "phi_L1", "ksi_L1", "du_L1", "dv_L1", "flow_u_add_L1",
                  "flow_u_med_L1", "flow_u_res_L0"):
            out["small_%s_%s" % (name, k)] = stages[k]

    # Gaussian taps and the level table (SURVEY 8c fixtures 4 and 5)
    for s in (0.45, 1.5):
        out["taps_%g" % s] = O.gaussian_taps(s)[0]
    shapes = [(584, 388, 0.9), (128, 128, 0.9), (1024, 1024, 0.5), (1920, 1080, 0.5), (4096, 4096, 0.5),
              (8192, 8192, 0.5)]
    out["level_table"] = np.array([[w, h, int(s * 100), O.max_warp_level(w, h, s)] for w, h, s in shapes])
    np.savez_compressed(os.path.join(HERE, "flow2d_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "flow2d_golden.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
