"""The fused strip kernel's two round-3 mechanisms against the oracle, bit for bit:
  * the three-step division (y = RN(1/den) once per pixel, q0 = n y, r = fma(-q0, den, n), q = fma(r, y, q0)) with its
    run-time guard and the repeat-with-plain-division fallback, on operands chosen to trip every guard;
  * the border-aware strip plan (shorter strips on the image borders, one-dimensional grid over the working blocks) on
    level sizes whose plans have many strips per column, ragged last strips and very few block columns."""
import numpy as np
import pytest

from conftest import in_container, level_fields

pytestmark = pytest.mark.gpu


def up(ctx, a, cw, ch):
    return ctx.plane(cw, ch, in_container(a, cw, ch))


def fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, cw, ch, hx, hy, alpha, outer, inner, constancy, algorithm=2):
    d = [up(ctx, a, cw, ch) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(6))
    rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, alpha, 0.001, 0.001, outer, inner, constancy,
                               algorithm)
    odu, odv, _, _ = oracle.solve_level(f0, f1, u, v, w, h, hx, hy, alpha, 0.001, 0.001, outer, inner, constancy)
    return rdu.download(w, h), rdv.download(w, h), odu, odv


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("constancy", [0, 1])
@pytest.mark.parametrize("w,h", [(640, 520), (300, 200)])
def test_ordinary_operands_do_not_fall_back(ctx, oracle, w, h, constancy):
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 11)
    before = ctx.fused_fallbacks()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(1.0), np.float32(1.0), 35.0, 3, 5,
                                     constancy)
    assert np.array_equal(a, odu) and np.array_equal(b, odv)
    assert ctx.fused_fallbacks() == before  # the three-step division served every wave


@pytest.mark.parametrize("inner", [5, 3])
def test_tiny_numerators_fall_back_to_the_plain_division(ctx, oracle, inner):
    """Flat frames (no data term) and a flow of magnitude 1e-30: every numerator of the point update is a sum of
    weight x (difference of 1e-30 numbers), far below 2^-80, where the residual of the three-step division leaves the
    normal range.  The guard sends every wave to the plain division and the result equals the oracle's."""
    w, h = 640, 260
    rng = np.random.default_rng(5)
    f0 = np.full((h, w), 100.0, np.float32)
    f1 = f0.copy()
    u = (rng.normal(0, 1, (h, w)) * 1e-30).astype(np.float32)
    v = (rng.normal(0, 1, (h, w)) * 1e-30).astype(np.float32)
    before = ctx.fused_fallbacks()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(1.0), np.float32(1.0), 35.0, 2, inner, 0)
    assert float(np.abs(odu).max()) > 0 and float(np.abs(odu).max()) < 1e-25
    assert np.array_equal(bits(a), bits(odu)) and np.array_equal(bits(b), bits(odv))
    assert ctx.fused_fallbacks() > before


def test_a_diffusion_front_trips_the_guard_only_where_it_is(ctx, oracle):
    """Flat frames and one bump in the flow: the increment spreads one pixel per sweep and decays by orders of magnitude
    per pixel, so the waves on the front see tiny non-zero numerators (fallback), the others exact zeros or ordinary
    numbers (no fallback).  Every pixel equals the oracle either way."""
    w, h = 1040, 700
    f0 = np.full((h, w), 50.0, np.float32)
    f1 = f0.copy()
    u = np.zeros((h, w), np.float32)
    v = np.zeros((h, w), np.float32)
    u[300:303, 500:503] = 1e-12
    before = ctx.fused_fallbacks()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(1.0), np.float32(1.0), 35.0, 8, 5, 0)
    assert np.array_equal(bits(a), bits(odu)) and np.array_equal(bits(b), bits(odv))
    tripped = ctx.fused_fallbacks() - before
    assert tripped > 0
    nonzero = np.abs(odu[odu != 0])
    assert nonzero.min() < 2.0 ** -80 < nonzero.max()  # the front did pass below the guard's threshold


def test_denominators_outside_the_proven_range_fall_back(ctx, oracle):
    """alpha = 1e22 puts the denominators (ksi J + sum of the face weights) near 1e25, above 2^40: fallback, same bits."""
    w, h = 640, 200
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 21)
    before = ctx.fused_fallbacks()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(1.0), np.float32(1.0), 1e22, 2, 5, 0)
    assert np.array_equal(bits(a), bits(odu)) and np.array_equal(bits(b), bits(odv))
    assert ctx.fused_fallbacks() > before


def test_overflowing_results_match_the_per_sweep_kernels(ctx, flow2d, oracle):
    """An infinite patch in the flow: the sweeps around it produce infinities and NaNs.  Non-finite results trip the
    output guard, and the repeat with the plain division delivers what the per-sweep kernels (one launch per reference
    launch) deliver: the same pixels NaN, the same infinite, every finite one identical."""
    w, h = 640, 200
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 23)
    u = u.copy()
    u[50:60, 100:400] = np.inf
    d = [up(ctx, a, w, h) for a in (f0, f1, u, v)]
    res = []
    before = ctx.fused_fallbacks()
    for algorithm in (flow2d.SOLVER_FUSED, flow2d.SOLVER_PER_SWEEP):
        du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h).fill_bytes(0) for _ in range(6))
        rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 2, 5, 0, algorithm)
        res.append((rdu.download(w, h), rdv.download(w, h)))
    assert not np.isfinite(res[1][0]).all()
    # NaN payloads are not part of the contract: compare finiteness class and every finite value
    for k in range(2):
        x, y = res[0][k], res[1][k]
        assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(np.isinf(x), np.isinf(y))
        fin = np.isfinite(y)
        assert np.array_equal(bits(x[fin]), bits(y[fin]))
    assert ctx.fused_fallbacks() > before


@pytest.mark.parametrize("constancy", [0, 1])
@pytest.mark.parametrize("w,h,cw,ch", [
    (1111, 777, 1120, 780),    # 6 block columns: border-aware plan with a ragged last middle strip
    (640, 1000, 640, 1000),    # 4 block columns, tall
    (2048, 1536, 2048, 1536),  # 10 block columns
    (417, 900, 448, 900),      # 3 block columns: one interior column only
    (416, 300, 416, 300),      # 2 block columns: uniform strips
    (1300, 64, 1312, 64),      # 7 block columns, level lower than two border strips
])
def test_border_aware_strip_plan_covers_every_pixel(ctx, oracle, w, h, cw, ch, constancy):
    """Poisoned output planes (0x7f bytes): a row or column no strip stores would differ from the oracle."""
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 31)
    hx, hy = np.float32(cw / w), np.float32(ch / h)
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, cw, ch, hx, hy, 3.5, 2, 5, constancy)
    assert np.array_equal(a, odu) and np.array_equal(b, odv)


def test_negative_zero_in_the_flow_falls_back(ctx, oracle):
    """A numerator of exactly -0 is the one zero the three-step division gets wrong (+0 where the quotient is -0).  It
    takes a -0 in the flow planes to make one: flat frames and a flow plane of -0 do (every face term is -0, the data
    term too).  The guard sees the -0 entries as they are read and sends the waves to the plain division: same bits as
    the oracle, signs of zeros included."""
    w, h = 640, 200
    f0 = np.full((h, w), 80.0, np.float32)
    f1 = f0.copy()
    u = np.full((h, w), -0.0, np.float32)
    v = np.zeros((h, w), np.float32)
    before = ctx.fused_fallbacks()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(1.0), np.float32(1.0), 35.0, 2, 5, 0)
    assert np.array_equal(bits(a), bits(odu)) and np.array_equal(bits(b), bits(odv))
    assert ctx.fused_fallbacks() > before
