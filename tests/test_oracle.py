"""CPU checks of the oracle itself (no GPU).

The reference ships no tests or golden vectors, so the oracle is pinned three ways:
 * against the anchor values SURVEY.md section 8(c) recorded from the reference's own sources run on
   rub1.raw / rub2.raw (test_rub_anchors_*),
 * against an independently written numpy restatement, bit for bit (test_cross_*),
 * against the level-count table of SURVEY.md section 8(a) row H2.
tests/data/rub{1,2}.raw are the reference's data files (8-bit 584x388), inputs only.
"""
import os

import numpy as np
import pytest

from conftest import level_fields

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def rub_pair():
    r1 = np.fromfile(os.path.join(DATA, "rub1.raw"), np.uint8).reshape(388, 584).astype(np.float32)
    r2 = np.fromfile(os.path.join(DATA, "rub2.raw"), np.uint8).reshape(388, 584).astype(np.float32)
    return r1, r2


def stats(a):
    return (a.mean(dtype=np.float64), a.std(dtype=np.float64), float(a.min()), float(a.max()))


def test_level_table(oracle):
    # GetMaxWarpLevel for the config shapes
    table = {(584, 388, 0.9): 47, (128, 128, 0.9): 36, (1024, 1024, 0.5): 9, (1920, 1080, 0.5): 9,
             (4096, 4096, 0.5): 11, (8192, 8192, 0.5): 12}
    for (w, h, s), want in table.items():
        assert oracle.max_warp_level(w, h, s) == want
    # rub, settings.xml: 20 levels, coarsest 79x53
    assert oracle.level_geometry(584, 388, 0.9, 19)[:2] == (79, 53)
    # scale >= 1 never enters the loop
    assert oracle.max_warp_level(100, 100, 1.0) == 0  # loop skipped, r_width == 1 -> decrement


def test_gaussian_taps(oracle):
    t, r = oracle.gaussian_taps(0.45)
    assert r == 1 and abs(float(t.sum()) - 1) < 1e-6
    t, r = oracle.gaussian_taps(1.5)
    assert r == 4 and len(t) == 9 and np.array_equal(t, t[::-1])


def test_rub_anchors_settings_xml(oracle):
    """settings.xml solver values; anchors from SURVEY.md 8(c) (reference source, no FP contraction)."""
    r1, r2 = rub_pair()
    u, v, _ = oracle.compute_flow(r1, r2, 20, 0.9, 20, 5, 3.5, 0.001, 0.001, 5, 0.45)
    for got, want in zip(stats(u), (0.04565, 1.12353, -3.6132, 2.5242)):
        assert abs(got - want) < 6e-5
    for got, want in zip(stats(v), (-0.11873, 0.43328, -3.6673, 2.1268)):
        assert abs(got - want) < 6e-5
    assert abs(float(u[194, 292]) - 1.246711) < 1e-6
    assert abs(float(v[194, 292]) - (-1.048284)) < 1e-6


def test_rub_anchors_main_defaults(oracle):
    """main.cpp defaults (levels 50 -> 47, outer 40, alpha 35, sigma 1.5); anchors from SURVEY.md 8(c)."""
    r1, r2 = rub_pair()
    u, v, _ = oracle.compute_flow(r1, r2, 50, 0.9, 40, 5, 35.0, 0.001, 0.001, 5, 1.5)
    for got, want in zip(stats(u), (0.29152, 0.69395, -1.5031, 2.2520)):
        assert abs(got - want) < 6e-5
    for got, want in zip(stats(v), (-0.17219, 0.22851, -1.1015, 0.6132)):
        assert abs(got - want) < 6e-5


@pytest.mark.parametrize("w,h", [(100, 70), (96, 64), (37, 20)])
def test_cross_kernels(oracle, w, h):
    from oracle import np_restatement as N
    f0, f1, u, v, du, dv = level_fields(oracle, w, h, 11)
    for s in (0.45, 1.5, 3.0):
        assert np.array_equal(oracle.convolution(f0, w, h, s), N.convolution(f0, s))
    for ow, oh in ((w * 4 // 5, h * 4 // 5), (13, 9), (5, 4), (w, h)):
        assert np.array_equal(oracle.resample(f0, w, h, ow, oh)[:oh, :ow], N.resample(f0, ow, oh))
    for hx, hy in ((1.0, 1.0), (1.25, 1.1)):
        assert np.array_equal(oracle.registration(f0, f1, u * 3, v * 3, w, h, hx, hy),
                              N.registration(f0, f1, u * 3, v * 3, hx, hy))
        phi, ksi = oracle.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, hx, hy, 0.001, 0.001)
        phi2, ksi2 = N.compute_phi_ksi(f0, f1, u, v, du, dv, hx, hy, 0.001, 0.001)
        assert np.array_equal(phi, phi2) and np.array_equal(ksi, ksi2)
        for g in (0, 1, 2):  # 2 = Gradient with true neighbours (the product's extra mode)
            a, b = oracle.solve_sweep(f0, f1, u, v, du, dv, phi, ksi, w, h, hx, hy, 35.0, g)
            a2, b2 = N.solve_sweep(f0, f1, u, v, du, dv, phi, ksi, hx, hy, 35.0, g)
            assert np.array_equal(a, a2) and np.array_equal(b, b2)
    for r in (3, 5, 7):
        assert np.array_equal(oracle.median(u, w, h, r), N.median(u, r))


def test_cross_upsample(oracle):
    from oracle import np_restatement as N
    f0, *_ = level_fields(oracle, 37, 20, 12)
    big = np.zeros((70, 100), np.float32)
    big[:20, :37] = f0
    assert np.array_equal(oracle.resample(big, 37, 20, 100, 70), N.resample(f0, 100, 70))


@pytest.mark.parametrize("gradient", [0, 1, 2])
def test_cross_end_to_end(oracle, gradient):
    from oracle import np_restatement as N
    f0, f1, *_ = level_fields(oracle, 100, 70, 13)
    uo, vo, _ = oracle.compute_flow(f0, f1, 6, 0.8, 2, 3, 3.5, 0.001, 0.001, 5, 0.45, gradient)
    un, vn = N.compute_flow(f0, f1, 6, 0.8, 2, 3, 3.5, 0.001, 0.001, 5, 0.45, gradient)
    assert np.array_equal(uo, un) and np.array_equal(vo, vn)


def test_median_matches_scipy(oracle):
    from scipy.ndimage import median_filter
    _, _, u, *_ = level_fields(oracle, 50, 31, 14)
    for r in (3, 5, 7):
        assert np.array_equal(oracle.median(u, 50, 31, r), median_filter(u, size=r, mode="mirror"))


def test_translation_is_recovered(oracle):
    """Sanity of the whole path: a translating pattern yields a flow close to the true shift."""
    f0, f1 = oracle.synthetic_pair(128, 96, 1.5, -0.75)
    u, v, _ = oracle.compute_flow(f0, f1, 12, 0.8, 10, 5, 3.5, 0.001, 0.001, 5, 0.45)
    inner = (slice(16, -16), slice(16, -16))
    assert abs(float(np.median(u[inner])) - 1.5) < 0.25
    assert abs(float(np.median(v[inner])) + 0.75) < 0.15


def test_sor_converges_to_the_jacobi_fixed_point_faster(oracle):
    """Property of the opt-in mode: for frozen robust weights, red-black SOR reaches the fixed point of the
    linear system that the reference's Jacobi sweeps approach, in far fewer iterations."""
    w, h = 48, 40
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 41)
    z = np.zeros_like(f0)
    phi, ksi = oracle.compute_phi_ksi(f0, f1, u * 0, v * 0, z, z, w, h, 1.0, 1.0, 0.5, 0.5)
    u0, v0 = u * 0, v * 0

    def jacobi(n):
        du, dv = z.copy(), z.copy()
        for _ in range(n):
            du, dv = oracle.solve_sweep(f0, f1, u0, v0, du, dv, phi, ksi, w, h, 1.0, 1.0, 3.5)
        return du, dv

    def sor(n, omega):
        du, dv = z.copy(), z.copy()
        for _ in range(n):
            du, dv = oracle.sor_iteration(f0, f1, u0, v0, du, dv, phi, ksi, w, h, 1.0, 1.0, 3.5, omega)
        return du, dv

    ref_du, ref_dv = jacobi(4000)
    err = lambda a: float(np.abs(a[0] - ref_du).max() + np.abs(a[1] - ref_dv).max())
    assert err(jacobi(8000)) < 1e-4                      # the long Jacobi run has converged
    assert err(sor(300, 1.7)) < 1e-3                     # SOR gets there in 300 iterations ...
    assert err(jacobi(300)) > 5 * err(sor(300, 1.7))     # ... where Jacobi is still far away
    assert err(sor(300, 1.0)) < err(jacobi(300))         # plain red-black Gauss-Seidel also beats Jacobi


# ---- the oracle against THE REFERENCE ITSELF -------------------------------------------------------------
# tests/golden/ref_kernels_golden.npz: outputs of the reference's own src/kernels/*_2d.cu, compiled for gfx950 from
# the sources where they lie and run on an MI355X (tests/golden/make_ref_golden.py).
# tests/golden/ref_host_golden.npz: outputs of the reference's own host sources run in the build container
# (tests/golden/make_ref_host_golden.py).  These are what pin the oracle.
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref_golden():
    return np.load(os.path.join(GOLDEN_DIR, "ref_kernels_golden.npz"))


@pytest.fixture(scope="module")
def ref_host_golden():
    import json
    g = np.load(os.path.join(GOLDEN_DIR, "ref_host_golden.npz"))
    return g, json.loads(str(g["meta"]))


def bits_equal(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                                                 np.ascontiguousarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("tag", ["odd", "tile"])
def test_ref_golden_kernels(oracle, ref_golden, tag):
    """Every reference kernel, on a 100x70 level in a 128x80 container and a 96x64 (16x8-multiple) level:
    the oracle's restatement gives the same bits."""
    g = ref_golden
    w, h, cw, ch = (int(x) for x in g[tag + "_geom"])
    hx, hy = (np.float32(x) for x in g[tag + "_h"])
    f0, f1, u, v, du, dv = (g["%s_in_%s" % (tag, k)] for k in ("f0", "f1", "u", "v", "du", "dv"))
    assert bits_equal(oracle.add(u, du, w, h), g[tag + "_add"])
    for sigma in (0.45, 1.5):
        assert bits_equal(oracle.convolution(f0, w, h, sigma), g["%s_conv_%g" % (tag, sigma)])
    m = g[tag + "_in_median_special"]  # NaN, +-0, ties: the reference's insertion sort with `<`
    for radius in (3, 5, 7):
        assert bits_equal(oracle.median(u, w, h, radius), g["%s_median_%d" % (tag, radius)])
        got, want = oracle.median(m, w, h, radius), g["%s_median_special_%d" % (tag, radius)]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert bits_equal(np.nan_to_num(got, nan=7.0), np.nan_to_num(want, nan=7.0))
    assert bits_equal(oracle.registration(f0, f1, g[tag + "_in_u_warp"], v, w, h, hx, hy), g[tag + "_registration"])
    n_resample = 0
    for key in g.files:
        if key.startswith(tag + "_resample_"):
            rw, rh = (int(x) for x in key.rsplit("_", 1)[1].split("x"))
            cont = np.zeros((max(ch, rh), max(cw, rw)), np.float32)
            cont[:h, :w] = f0
            assert bits_equal(oracle.resample(cont, w, h, rw, rh)[:rh, :rw], g[key]), key
            n_resample += 1
    assert n_resample >= 3
    phi, ksi = oracle.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, hx, hy, 0.001, 0.001)
    assert bits_equal(phi, g[tag + "_phi"]) and bits_equal(ksi, g[tag + "_ksi"])
    modes = [("grey", oracle.GREY)] + ([("grad", oracle.GRADIENT)] if tag == "tile" else [])
    for name, c in modes:
        a, b = oracle.solve_sweep(f0, f1, u, v, du, dv, phi, ksi, w, h, hx, hy, 35.0, c)
        assert bits_equal(a, g["%s_sweep_%s_du" % (tag, name)]) and bits_equal(b, g["%s_sweep_%s_dv" % (tag, name)])
        sdu, sdv, sphi, _ = oracle.solve_level(f0, f1, u, v, w, h, hx, hy, 35.0, 0.001, 0.001, 3, 5, c)
        assert bits_equal(sdu, g["%s_solve_%s_du" % (tag, name)]) and bits_equal(sdv, g["%s_solve_%s_dv" % (tag, name)])
        assert bits_equal(sphi, g["%s_solve_%s_phi" % (tag, name)])


def test_ref_golden_gradient_off_the_tile_grid(oracle, ref_golden):
    """solve_2d_grad on a 100x70 level: the image edge falls inside a 16x8 block, where the reference's last row
    and column read a shared-memory slot no thread wrote (solve_2d.cu:795-842,872-876).  Everywhere else one sweep
    is bit-identical; on that row and column the reference's value is whatever the slot held."""
    g = ref_golden
    w, h = 100, 70
    hx, hy = (np.float32(x) for x in g["odd_h"])
    f0, f1, u, v, du, dv = (g["odd_in_%s" % k] for k in ("f0", "f1", "u", "v", "du", "dv"))
    a, b = oracle.solve_sweep(f0, f1, u, v, du, dv, g["odd_phi"], g["odd_ksi"], w, h, hx, hy, 35.0, oracle.GRADIENT)
    assert bits_equal(a[:-1, :-1], g["odd_sweep_grad_du"][:-1, :-1])
    assert bits_equal(b[:-1, :-1], g["odd_sweep_grad_dv"][:-1, :-1])


def test_ref_golden_log_derivatives(oracle, ref_golden):
    """solve_2d_log: same structure as the reference's kernel; logf comes from the CPU's libm here and from the
    device library there, so the comparison allows last-place differences of log(I + 1) (most pixels are
    bit-identical, the rest differ by a few ulp of the result)."""
    g = ref_golden
    w, h = 96, 64
    hx, hy = (np.float32(x) for x in g["tile_h"])
    f0, f1, u, v, du, dv = (g["tile_in_%s" % k] for k in ("f0", "f1", "u", "v", "du", "dv"))
    a, b = oracle.solve_sweep(f0, f1, u, v, du, dv, g["tile_phi"], g["tile_ksi"], w, h, hx, hy, 35.0,
                              oracle.LOG_DERIVATIVES)
    for got, want in ((a, g["tile_sweep_log_du"]), (b, g["tile_sweep_log_dv"])):
        assert float(np.abs(got - want).max()) < 2e-6 and float((got == want).mean()) > 0.98
    sdu, sdv, _, _ = oracle.solve_level(f0, f1, u, v, w, h, hx, hy, 35.0, 0.001, 0.001, 3, 5, oracle.LOG_DERIVATIVES)
    assert float(np.abs(sdu - g["tile_solve_log_du"]).max()) < 1e-5
    assert float(np.abs(sdv - g["tile_solve_log_dv"]).max()) < 1e-5


FLOW_RUNS = ["rub_short", "rub_settings", "rub_main_defaults", "syn_grey", "syn_grad", "odd_grey"]


@pytest.mark.parametrize("name", FLOW_RUNS)
def test_ref_golden_compute_flow(oracle, ref_golden, name):
    """Whole ComputeFlow runs of the reference's kernels (rub1/rub2 with the settings.xml values and with the
    main.cpp defaults, synthetic pairs in Grey and Gradient mode): sha256 of the full u and v fields."""
    import hashlib
    g = ref_golden
    if name.startswith("rub"):
        a, b = rub_pair()
    elif name.startswith("syn"):
        a, b = g["syn_f0"], g["syn_f1"]
    else:
        a, b = g["odd_f0"], g["odd_f1"]
    p = g[name + "_params"]
    u, v, _ = oracle.compute_flow(a, b, int(p[0]), float(p[1]), int(p[2]), int(p[3]), float(p[4]), float(p[5]),
                                  float(p[6]), int(p[7]), float(p[8]), int(p[9]))
    sha = lambda x: hashlib.sha256(np.ascontiguousarray(x, np.float32).tobytes()).hexdigest()
    sub = {"rub_short": 2, "rub_settings": 4, "rub_main_defaults": 4}.get(name, 1)
    assert bits_equal(u[::sub, ::sub], g[name + "_u"]) and bits_equal(v[::sub, ::sub], g[name + "_v"])
    assert [sha(u), sha(v)] == list(g[name + "_sha"])


@pytest.mark.parametrize("name,gate", [("syn_log", 2e-3), ("syn_log_b", 1e-4)])
def test_ref_golden_compute_flow_log(oracle, ref_golden, name, gate):
    """LogDerivatives through a whole pyramid.  The reference's kernel takes logf from the GPU's device library (built
    on the hardware's log2 instruction), the oracle from the CPU's libm; they differ in the last place for some
    inputs, and a pyramid amplifies that by the conditioning of the run (alpha 0.0005: RMSE ~5e-4 on a 0.9 px flow;
    alpha 0.02: below the 1e-4 gate).  Bit-exactness in this mode is asserted where it is meaningful: HIP path vs
    the reference's kernels on the same GPU (tests/test_gpu_reference.py)."""
    g = ref_golden
    p = g[name + "_params"]
    u, v, _ = oracle.compute_flow(g["syn_f0"], g["syn_f1"], int(p[0]), float(p[1]), int(p[2]), int(p[3]), float(p[4]),
                                  float(p[5]), float(p[6]), int(p[7]), float(p[8]), oracle.LOG_DERIVATIVES)
    assert float(np.abs(g[name + "_u"]).max()) > 0.01  # a real flow, not a field smoothed to zero
    rmse = lambda x, y: float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2)))
    assert rmse(u, g[name + "_u"]) < gate and rmse(v, g[name + "_v"]) < gate


def test_ref_host_golden_levels_and_taps(oracle, ref_host_golden):
    """GetMaxWarpLevel and ComputeGaussianKernel of the reference's own host code."""
    _, meta = ref_host_golden
    for w, h, s, want in meta["levels"]:
        assert oracle.max_warp_level(w, h, s) == want, (w, h, s)
    for sigma, rec in meta["taps"].items():
        taps, r = oracle.gaussian_taps(float(sigma))
        assert r == rec["radius"]
        assert ["%08x" % b for b in taps.view(np.uint32)] == rec["bits"], sigma
