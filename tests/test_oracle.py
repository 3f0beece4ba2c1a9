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
