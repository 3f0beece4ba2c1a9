"""Generates tests/golden/ref_kernels_golden.npz by RUNNING THE REFERENCE'S OWN KERNELS on an MI355X.

oracle/_ref/*.co are the reference's src/kernels/*_2d.cu compiled for gfx950 from the sources where they
lie (oracle/Makefile, -ffp-contract=off); oracle/ref_driver.cpp launches them by symbol name with the
geometry of the reference's operator layer.  This script feeds them seeded inputs and records inputs and
outputs, per kernel and for whole ComputeFlow runs.  The fixture is data (inputs and expected outputs);
the CPU tests (tests/test_oracle.py::test_ref_golden_*) hold the oracle to it bit for bit, which is what
pins the oracle to the reference.

Run on the GPU box:   python tests/golden/make_ref_golden.py            (writes gpurun_out/ref_kernels_golden.npz)
then copy the file to tests/golden/.  Prints, for information, whether the oracle agrees already.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402
from oracle import ref_kernels as RK  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()


def level_fields(w, h, seed):
    rng = np.random.default_rng(seed)
    f0, f1 = O.synthetic_pair(w, h, 1.5, -0.75)
    f1 = (f1 + rng.uniform(-1, 1, f1.shape)).astype(np.float32)
    u = rng.normal(0, 1, (h, w)).astype(np.float32)
    v = rng.normal(0, 1, (h, w)).astype(np.float32)
    du = rng.normal(0, 0.3, (h, w)).astype(np.float32)
    dv = rng.normal(0, 0.3, (h, w)).astype(np.float32)
    return f0, f1, u, v, du, dv


def rub():
    d = os.path.join(os.path.dirname(HERE), "data")
    r1 = np.fromfile(os.path.join(d, "rub1.raw"), np.uint8).reshape(388, 584).astype(np.float32)
    r2 = np.fromfile(os.path.join(d, "rub2.raw"), np.uint8).reshape(388, 584).astype(np.float32)
    return r1, r2


def report(name, got, want):
    same = np.array_equal(got, want, equal_nan=True)
    extra = "" if same else "  MISMATCH: %d px, max |d| %g" % (int((got != want).sum()), float(np.nanmax(np.abs(got - want))))
    print("%-44s %s%s" % (name, "oracle ==" if same else "oracle !=", extra))
    return same


def per_kernel(out, tag, w, h, cw, ch, seed, tile_multiple):
    f0, f1, u, v, du, dv = level_fields(w, h, seed)
    hx, hy = np.float32(1.25), np.float32(1.1)
    for k, a in zip(("f0", "f1", "u", "v", "du", "dv"), (f0, f1, u, v, du, dv)):
        out["%s_in_%s" % (tag, k)] = a
    out["%s_geom" % tag] = np.array([w, h, cw, ch], np.int64)
    out["%s_h" % tag] = np.array([hx, hy], np.float32)
    with RK.RefKernels(cw, ch) as R:
        got, _ = R.add(u, du)
        out["%s_add" % tag] = got
        report(tag + " add_2d", got, O.add(u, du, w, h))
        for sigma in (0.45, 1.5):
            got = R.convolution(f0, sigma)
            out["%s_conv_%g" % (tag, sigma)] = got
            report(tag + " convolution sigma %g" % sigma, got, O.convolution(f0, w, h, sigma))
            taps, r = R.gaussian_taps(sigma)
            report(tag + " taps %g" % sigma, taps, O.gaussian_taps(sigma)[0])
        for radius in (3, 5, 7):
            rc, got, _ = R.median(u, radius)
            out["%s_median_%d" % (tag, radius)] = got
            report(tag + " median_2d %d" % radius, got, O.median(u, w, h, radius))
        # NaN, +-0 and ties through the median (median_2d.cu:52-63 insertion sort with `<`)
        m = u.copy()
        m[::3, ::5] = 0.0
        m[1::4, 2::7] = -0.0
        m[5, 7] = np.nan
        m[h // 2, w // 2:w // 2 + 3] = np.nan
        m[h - 1, w - 1] = np.nan
        out["%s_in_median_special" % tag] = m
        for radius in (3, 5, 7):
            rc, got, _ = R.median(m, radius)
            out["%s_median_special_%d" % (tag, radius)] = got
            want = O.median(m, w, h, radius)
            bits_same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
            report(tag + " median_2d %d NaN/+-0 (bits %s)" % (radius, bits_same), got, want)
        uu = u * 4
        uu[0, 0] = np.nan
        uu[h - 1, w - 1] = 1e9
        uu[h // 2, w // 2] = -1e9
        out["%s_in_u_warp" % tag] = uu
        got = R.registration(f0, f1, uu, v, hx, hy)
        out["%s_registration" % tag] = got
        report(tag + " registration_2d", got, O.registration(f0, f1, uu, v, w, h, hx, hy))
        for rw, rh in ((w * 4 // 5, h * 4 // 5), (w // 3 + 1, h // 4 + 1), (w + 7, h + 5), (5, 4)):
            if rw > cw or rh > ch:
                continue
            got = R.resample(f0, rw, rh)
            out["%s_resample_%dx%d" % (tag, rw, rh)] = got
            cont = np.zeros((ch, cw), np.float32)
            cont[:h, :w] = f0
            report(tag + " resample -> %dx%d" % (rw, rh), got, O.resample(cont, w, h, rw, rh)[:rh, :rw])
        phi, ksi = R.phi_ksi(f0, f1, u, v, du, dv, hx, hy, 0.001, 0.001)
        out["%s_phi" % tag], out["%s_ksi" % tag] = phi, ksi
        ophi, oksi = O.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, hx, hy, 0.001, 0.001)
        report(tag + " compute_phi_ksi phi", phi, ophi)
        report(tag + " compute_phi_ksi ksi", ksi, oksi)
        for cname, c in (("grey", RK.GREY), ("grad", RK.GRADIENT), ("log", RK.LOG_DERIVATIVES)):
            tdu, tdv = R.sweep(c, f0, f1, u, v, du, dv, phi, ksi, hx, hy, 35.0)
            out["%s_sweep_%s_du" % (tag, cname)], out["%s_sweep_%s_dv" % (tag, cname)] = tdu, tdv
            oc = {RK.GREY: O.GREY, RK.GRADIENT: O.GRADIENT, RK.LOG_DERIVATIVES: O.LOG_DERIVATIVES}[c]
            if True:
                odu, odv = O.solve_sweep(f0, f1, u, v, du, dv, phi, ksi, w, h, hx, hy, 35.0, oc)
                ok = report(tag + " solve_2d[%s] du" % cname, tdu, odu)
                report(tag + " solve_2d[%s] dv" % cname, tdv, odv)
                if not ok and not tile_multiple:
                    bad = np.argwhere(tdu != odu)
                    print("    mismatching x:", sorted(set(bad[:, 1].tolist()))[:20], " y:", sorted(set(bad[:, 0].tolist()))[:20])
            # second launch: is the result reproducible (the Gradient/Log kernels read unwritten LDS off the tile grid)?
            tdu2, _ = R.sweep(c, f0, f1, u, v, du, dv, phi, ksi, hx, hy, 35.0)
            print("%-44s %s" % (tag + " solve_2d[%s] relaunch" % cname,
                                "identical" if np.array_equal(tdu, tdu2, equal_nan=True) else "DIFFERS"))
            sdu, sdv, sphi, sksi, _ = R.solve(f0, f1, u, v, hx, hy, c, 3, 5, 35.0, 0.001, 0.001)
            out["%s_solve_%s_du" % (tag, cname)], out["%s_solve_%s_dv" % (tag, cname)] = sdu, sdv
            out["%s_solve_%s_phi" % (tag, cname)] = sphi
            if True:
                odu, odv, ophi2, _ = O.solve_level(f0, f1, u, v, w, h, hx, hy, 35.0, 0.001, 0.001, 3, 5, oc)
                report(tag + " Execute 3x5 [%s] du" % cname, sdu, odu)
                report(tag + " Execute 3x5 [%s] dv" % cname, sdv, odv)


def flows(out):
    r1, r2 = rub()
    runs = [
        ("rub_short", r1, r2, (8, 0.8, 3, 5, 3.5, 0.001, 0.001, 5, 0.45), RK.GREY, 2),
        ("rub_settings", r1, r2, (20, 0.9, 20, 5, 3.5, 0.001, 0.001, 5, 0.45), RK.GREY, 4),
        ("rub_main_defaults", r1, r2, (50, 0.9, 40, 5, 35.0, 0.001, 0.001, 5, 1.5), RK.GREY, 4),
    ]
    f0, f1 = O.synthetic_pair(256, 128, 1.5, -0.75, seed=1, noise=True)
    out["syn_f0"], out["syn_f1"] = f0, f1
    for cname, c in (("grey", RK.GREY), ("grad", RK.GRADIENT), ("log", RK.LOG_DERIVATIVES)):
        # log derivatives are ~1/I of the grey ones: a small alpha, or the flow is smoothed to nothing
        alpha = 0.0005 if c == RK.LOG_DERIVATIVES else 35.0
        runs.append(("syn_" + cname, f0, f1, (4, 0.5, 3, 5, alpha, 0.001, 0.001, 5, 1.5), c, 1))
    # a second LogDerivatives run, better conditioned (a CPU libm's logf differs from the GPU's in the last place;
    # the smaller alpha is, the more a pyramid amplifies that)
    runs.append(("syn_log_b", f0, f1, (3, 0.5, 3, 5, 0.02, 0.001, 0.001, 5, 1.5), RK.LOG_DERIVATIVES, 1))
    g0, g1 = O.synthetic_pair(100, 70, 1.5, -0.75, seed=2, noise=True)
    out["odd_f0"], out["odd_f1"] = g0, g1
    runs.append(("odd_grey", g0, g1, (6, 0.8, 2, 3, 3.5, 0.001, 0.001, 3, 0.45), RK.GREY, 1))
    for name, a, b, p, c, sub in runs:
        with RK.RefKernels(a.shape[1], a.shape[0]) as R:
            u, v, total, finest = R.compute_flow(a, b, *p, constancy=c)
        out[name + "_params"] = np.array(list(p) + [c], np.float64)
        out[name + "_u"], out[name + "_v"] = u[::sub, ::sub], v[::sub, ::sub]
        out[name + "_sha"] = np.array([sha(u), sha(v)])
        print("%-20s reference kernels: %.1f ms total, finest-level solve %.2f ms" % (name, total, finest))
        oc = {RK.GREY: O.GREY, RK.GRADIENT: O.GRADIENT, RK.LOG_DERIVATIVES: O.LOG_DERIVATIVES}[c]
        ou, ov, _ = O.compute_flow(a, b, *p, oc)
        report(name + " ComputeFlow u", u, ou)
        report(name + " ComputeFlow v", v, ov)
        if name == "rub_settings":
            with RK.RefKernels(a.shape[1], a.shape[0], fma=True) as R:
                fu, fv, _, _ = R.compute_flow(a, b, *p, constancy=c)
            rm = lambda x, y: float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2)))
            print("  same sources with hipcc's default FMA contraction: RMSE u %.3g v %.3g, max |du| %.3g" %
                  (rm(fu, u), rm(fv, v), float(np.abs(fu - u).max())))
            out["rub_settings_fma_rmse"] = np.array([rm(fu, u), rm(fv, v)])


def main():
    out = {}
    per_kernel(out, "odd", 100, 70, 128, 80, 11, False)   # not a multiple of any tile
    per_kernel(out, "tile", 96, 64, 96, 64, 12, True)     # multiple of 16 x 8: Gradient/Log are defined
    flows(out)
    dst = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dst, exist_ok=True)
    path = os.path.join(dst, "ref_kernels_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
