"""The fused strip kernel's round-3 mechanisms against the oracle, bit for bit:
  * the short reciprocal (v_rcp + one fma pair) and the short 1 / (2 sqrt(s)) of the coefficient stage, whose arguments
    the same guard watches (2 sqrt(s) and the denominators within [2^-30, 2^40]);
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


def fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, cw, ch, hx, hy, alpha, outer, inner, constancy, algorithm=2,
                    e_smooth=0.001, e_data=0.001):
    d = [up(ctx, a, cw, ch) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(6))
    rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, alpha, e_smooth, e_data, outer, inner, constancy,
                               algorithm)
    odu, odv, _, _ = oracle.solve_level(f0, f1, u, v, w, h, hx, hy, alpha, e_smooth, e_data, outer, inner, constancy)
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


@pytest.mark.parametrize("alpha,fuses", [(1e-29, True), (1e-31, False), (0.0, True), (3e-36, False)])
def test_neighbour_weights_the_strips_cannot_halve_exactly(ctx, flow2d, oracle, alpha, fuses):
    """Stage W multiplies (phi_n + phi_c) by HALF the neighbour weight alpha / h^2 (round 5) -- the same bits as (phi_n + phi_c) / 2
    times the weight while halving the weight is exact.  Weights below 2^-100 are not the strips' business: AUTO gives the level to
    the per-sweep kernels, an explicit request is refused; every path agrees with the oracle."""
    w, h = 1300, 300  # (above AUTO's tile threshold: the strips' level)
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 27)
    hx = hy = np.float32(1.0)
    same = lambda x, y: np.array_equal(x, y, equal_nan=True)  # (alpha = 0 leaves 0 / 0 where the image is flat: NaN on both sides)
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, hx, hy, alpha, 2, 5, 0, algorithm=0)
    assert same(a, odu) and same(b, odv)
    if fuses:
        a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, hx, hy, alpha, 2, 5, 0, algorithm=2)
        assert same(a, odu) and same(b, odv)
    else:
        with pytest.raises(flow2d.Flow2DError) as e:
            fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, hx, hy, alpha, 2, 5, 0, algorithm=2)
        assert e.value.status == 5


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


@pytest.mark.parametrize("constancy", [0, 1])
@pytest.mark.parametrize("e_smooth,e_data,scale,trips", [
    (1e-8, 1e-8, 1.0, False),     # 2 sqrt(s) down to 2e-8 = 2^-25.6: still inside the range
    (1e-20, 0.001, 1.0, True),    # e_smooth^2 = 1e-40 is a denormal: s of a flat flow region is, too
    (0.001, 1e-20, 1.0, True),    # the same for the data term where the frames agree
    (0.001, 0.001, 1e16, True),   # frames of magnitude 1e18: the data term's argument is far above 2^78
])
def test_robustifier_arguments_outside_the_proven_range_fall_back(ctx, oracle, constancy, e_smooth, e_data, scale, trips):
    """phi and ksi are 1 / (2 sqrt(s)) through the hardware root and reciprocal plus residual steps, proven for every s
    with 2 sqrt(s) in [2^-30, 2^40] (tools/ubench/rcp_sqrt_exhaustive.hip).  Regularisers so small that s is a denormal
    on flat regions, or frames so large that s overflows the range, trip the guard; either way every bit is the oracle's."""
    w, h = 640, 200
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 41)
    f0, f1 = (f0 * np.float32(scale)).astype(np.float32), (f1 * np.float32(scale)).astype(np.float32)
    u, v = u.copy(), v.copy()
    u[40:120, 100:500] = 0.25  # a flat flow region: the smoothness argument is e_smooth^2 there
    v[40:120, 100:500] = -0.5
    f1[60:100, 200:400] = f0[60:100, 200:400]  # frames agree, no gradient mismatch: ft = 0
    before = ctx.fused_fallbacks()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(1.0), np.float32(1.0), 35.0, 2, 5,
                                     constancy, e_smooth=e_smooth, e_data=e_data)
    same = np.array_equal(np.isnan(a), np.isnan(odu)) and np.array_equal(np.isnan(b), np.isnan(odv))
    fin_a, fin_b = ~np.isnan(odu), ~np.isnan(odv)
    assert same and np.array_equal(bits(a[fin_a]), bits(odu[fin_a])) and np.array_equal(bits(b[fin_b]), bits(odv[fin_b]))
    if trips:
        assert ctx.fused_fallbacks() > before


def test_zero_regularisers_match_the_per_sweep_kernels(ctx, flow2d, oracle):
    """e_smooth = e_data = 0 on a flat flow over agreeing frames: both arguments are exactly 0, the reference's phi and
    ksi are infinite and the sweeps produce NaNs.  The short forms give NaN for a zero argument, which trips the guard;
    the repeat with the plain expressions delivers what the per-sweep kernels deliver."""
    w, h = 640, 200
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 43)
    u, v = u.copy(), v.copy()
    u[40:120, 100:500] = 0.25
    v[40:120, 100:500] = -0.5
    f1 = f1.copy()
    f1[60:100, 200:400] = f0[60:100, 200:400]
    d = [up(ctx, a, w, h) for a in (f0, f1, u, v)]
    res = []
    before = ctx.fused_fallbacks()
    for algorithm in (flow2d.SOLVER_FUSED, flow2d.SOLVER_PER_SWEEP):
        du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h).fill_bytes(0) for _ in range(6))
        rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.0, 0.0, 2, 5, 0, algorithm)
        res.append((rdu.download(w, h), rdv.download(w, h)))
    assert not np.isfinite(res[1][0]).all()
    for k in range(2):
        x, y = res[0][k], res[1][k]
        assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(np.isinf(x), np.isinf(y))
        fin = np.isfinite(y)
        assert np.array_equal(bits(x[fin]), bits(y[fin]))
    assert ctx.fused_fallbacks() > before


@pytest.mark.parametrize("constancy", [0, 1, 2])  # (solve_2d_log with such spacings: test_gpu_reference.py, against the reference's kernel)
@pytest.mark.parametrize("hx,hy", [(1.1, 0.9), (3.7, 1.0), (0.013, 0.02)])
def test_spacings_that_are_no_powers_of_two(ctx, oracle, hx, hy, constancy):
    """The six divisions by 2h and 4h of a row step go through the three-step division with the host's RN(1 / (2h)),
    RN(1 / (4h)) when the spacing is no power of two (pyramids with a scale factor other than 0.5); same bits as the
    oracle's plain divisions, and ordinary operands do not fall back."""
    w, h = 640, 264
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 51)
    before = ctx.fused_fallbacks()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(hx), np.float32(hy), 35.0, 2, 5, constancy)
    assert np.array_equal(bits(a), bits(odu)) and np.array_equal(bits(b), bits(odv))
    if hx >= 1.0:  # (h = 0.013 makes second derivatives of 1e5 and more: some denominators leave the guarded range)
        assert ctx.fused_fallbacks() == before


def test_tiny_differences_over_a_non_power_of_two_spacing_fall_back(ctx, oracle):
    """A flow of magnitude 1e-30 over h = 1.1: the numerators of the flow derivatives (differences of such values) are
    below 2^-80, where the three-step division by 2h is not proven -- the guard sends the waves to the plain division."""
    w, h = 640, 200
    rng = np.random.default_rng(7)
    f0, f1, _, _, _, _ = level_fields(oracle, w, h, 53)
    u = (rng.normal(0, 1, (h, w)) * 1e-30).astype(np.float32)
    v = (rng.normal(0, 1, (h, w)) * 1e-30).astype(np.float32)
    before = ctx.fused_fallbacks()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(1.1), np.float32(1.1), 35.0, 1, 5, 0)
    assert np.array_equal(bits(a), bits(odu)) and np.array_equal(bits(b), bits(odv))
    assert ctx.fused_fallbacks() > before


def test_a_spacing_outside_the_guarded_range_takes_the_plain_divisions(ctx, oracle):
    """h = 3 * 2^40: 2h is above 2^40, outside the range the three-step division is proven for; the launch runs the
    plain pass throughout -- every wave counted as a plain-only wave, none as a guard trip -- and equals the oracle."""
    w, h = 640, 200
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 55)
    big = np.float32(3.0 * 2.0 ** 40)
    before, plain_before = ctx.fused_fallbacks(), ctx.fused_plain_waves()
    a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, big, np.float32(1.1), 35.0, 1, 5, 0)
    assert np.array_equal(bits(a), bits(odu)) and np.array_equal(bits(b), bits(odv))
    assert ctx.fused_plain_waves() > plain_before
    assert ctx.fused_fallbacks() == before  # the guard counter is for guard trips only


@pytest.mark.parametrize("inner", [1, 2, 5])
def test_launch_order_is_a_permutation_of_the_plan(ctx, inner):
    """Round 6: every XCD takes a contiguous run of the plan's blocks AND its share of the side blocks (first / last block column) of a
    border-aware plan, so that the eight XCDs carry equal work.  Whatever the geometry, the launch order must name every block of
    the plan exactly once, the strips of a block column must tile the level's rows, and the side blocks must be spread over the runs."""
    valid = 64 - 2 * (inner + 1)
    for w, h, inst in [(4096, 4096, 1), (2048, 2048, 1), (1024, 1024, 1), (1920, 1080, 8), (584, 388, 32), (8192, 8192, 1), (640, 520, 1),
                       (300, 200, 1), (4096, 2048, 2), (700, 4000, 1), (5000, 333, 3), (97, 61, 1), (2049, 1023, 1)]:
        order = ctx.fused_block_order(w, h, inner, inst)
        assert len(order) % 8 == 0
        live = order[order[:, 0] >= 0]
        blocks = set(map(tuple, live[:, :2].tolist()))
        assert len(blocks) == len(live), (w, h, inst, "a block named twice")
        blocks_x = (-(-w // valid) + 3) // 4
        assert {bx for bx, _ in blocks} == set(range(blocks_x)), (w, h, inst)
        for bx in range(blocks_x):
            rows = sorted((y0, y1) for cx, _, y0, y1 in live.tolist() if cx == bx and y1 > y0)
            assert rows[0][0] == 0 and rows[-1][1] == h, (w, h, inst, bx, rows[:2], rows[-2:])
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:])), (w, h, inst, bx)
        # the runs of the eight XCDs (launch ids k, k + 8, ...) carry the side blocks evenly (the 4096^2 plan: 62 of them, 7 or 8 per run)
        if (w, h, inst, inner) == (4096, 4096, 1, 5):
            side = [int(((order[k::8, 0] == 0) | (order[k::8, 0] == blocks_x - 1)).sum()) for k in range(8)]
            assert sum(side) == 62 and max(side) - min(side) <= 1, side


def test_clock_probe_reports_a_plausible_clock_per_xcd(flow2d, ctx):
    """flow2d_clock_probe_start / _read: one sleeping wave per XCD brackets a stretch of time with the constant 100 MHz clock and the
    shader clock -- the figure bench.py puts beside the strip kernel's launch time (the chip holds 1.6-2.4 GHz depending on the
    power the running kernels draw).  Alone on the device the probe sees the idle-to-boost range on every XCD a wave landed on."""
    probe = flow2d.Context(0)
    try:
        probe.clock_probe_start(300.0)
        ghz = probe.clock_probe_read()
    finally:
        probe.close()
    seen = [g for g in ghz if g > 0]
    assert len(ghz) == 8 and len(seen) >= 4, ghz       # eight workgroups dealt over the XCDs
    assert all(0.1 < g < 3.0 for g in seen), ghz
    with pytest.raises(flow2d.Flow2DError):
        ctx.clock_probe_start(0.0)


@pytest.mark.parametrize("constancy", [0, 1])
def test_lone_context_uses_the_packed_build_with_the_same_bits(flow2d, ctx, oracle, constancy):
    """flow2d_context_set_lone: a strip launch of at most one workgroup per CU takes the build of the strip kernel with packed
    arithmetic (no partner wave to share issue turns with) -- the same IEEE operations, so the same bits as the pipeline's build
    and the oracle, including a launch that continues an outer iteration's sweeps (no packed kernels for those: falls through)."""
    w, h = 1024, 512   # 20 strips wide, few strips per column: under one workgroup per CU
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 21)
    results = []
    for lone in (False, True):
        ctx.set_lone(lone)
        for inner in (5, 8):
            a, b, odu, odv = fused_vs_oracle(ctx, oracle, f0, f1, u, v, w, h, w, h, np.float32(1.0), np.float32(1.0), 35.0, 2, inner, constancy)
            assert np.array_equal(bits(a), bits(odu)) and np.array_equal(bits(b), bits(odv)), (lone, inner)
    ctx.set_lone(False)
