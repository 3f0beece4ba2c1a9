"""Per-kernel parity of the HIP path (through the C-ABI) against the CPU oracle: bit-exact.

Each case places the level in the top-left corner of a larger container, like the reference's
full-resolution pitched planes, and uses sizes that are not tile multiples.
"""
import numpy as np
import pytest

from conftest import in_container, level_fields

pytestmark = pytest.mark.gpu

# (level w, level h, container w, container h)
SIZES = [(100, 70, 128, 80), (64, 64, 64, 64), (5, 4, 40, 30), (257, 33, 300, 40), (16, 8, 16, 8)]


def up(ctx, a, cw, ch, fill=0.0):
    return ctx.plane(cw, ch, in_container(a, cw, ch, fill))


@pytest.mark.parametrize("w,h,cw,ch", SIZES)
def test_add(ctx, oracle, w, h, cw, ch):
    f0, f1, *_ = level_fields(oracle, w, h, 1)
    a, b = up(ctx, f0, cw, ch, 7.0), up(ctx, f1, cw, ch, 9.0)
    ctx.add(a, b, w, h)
    got = a.download()
    assert np.array_equal(got[:h, :w], oracle.add(f0, f1, w, h))
    # nothing outside the level rectangle is touched
    assert np.all(got[h:, :] == 7.0) and np.all(got[:, w:] == 7.0)


@pytest.mark.parametrize("sigma", [0.45, 0.7, 1.0, 1.5, 1.9, 2.2, 3.0, 8.3])  # radii 1..6 stream, larger ones tile
@pytest.mark.parametrize("w,h,cw,ch", SIZES)
def test_gaussian(ctx, flow2d, oracle, w, h, cw, ch, sigma):
    f0, *_ = level_fields(oracle, w, h, 2)
    taps, r = flow2d.gaussian_kernel(sigma)
    otaps, orad = oracle.gaussian_taps(sigma)
    assert r == orad and np.array_equal(taps, otaps)
    src, tmp, dst = up(ctx, f0, cw, ch, 5.0), ctx.plane(cw, ch), ctx.plane(cw, ch)
    ctx.convolution_rows(tmp, src, w, h, taps, r)
    ctx.convolution_columns(dst, tmp, w, h, taps, r)
    want = oracle.convolution(f0, w, h, sigma)
    assert np.array_equal(dst.download(w, h), want)
    fused = ctx.plane(cw, ch).fill_bytes(0x7f)  # both passes in one launch
    ctx.gaussian_blur(fused, src, w, h, taps, r)
    assert np.array_equal(fused.download(w, h), want)


@pytest.mark.parametrize("sigma", [0.45, 1.5, 2.2])
@pytest.mark.parametrize("w,h,cw,ch", [(700, 133, 704, 140), (1000, 300, 1024, 300), (57, 200, 64, 200)])
def test_gaussian_streaming_strips(ctx, flow2d, oracle, w, h, cw, ch, sigma):
    """Several column strips and row strips of the one-launch blur; container larger than the level."""
    f0, *_ = level_fields(oracle, w, h, 12)
    taps, r = flow2d.gaussian_kernel(sigma)
    src, dst = up(ctx, f0, cw, ch, 5.0), ctx.plane(cw, ch).fill_bytes(0x7f)
    ctx.gaussian_blur(dst, src, w, h, taps, r)
    assert np.array_equal(dst.download(w, h), oracle.convolution(f0, w, h, sigma))


@pytest.mark.parametrize("w,h,ow,oh", [(100, 70, 80, 64), (100, 70, 37, 20), (100, 70, 13, 9), (100, 70, 5, 4),
                                       (37, 20, 100, 70), (64, 64, 32, 32), (33, 17, 34, 18), (1000, 37, 77, 9), (4096, 8, 32, 8),
                                       (5000, 5, 2, 3), (300, 40, 150, 40), (257, 13, 128, 6)])
def test_resample(ctx, oracle, w, h, ow, oh):
    cw, ch = max(w, ow) + 3, max(h, oh) + 2
    f0, *_ = level_fields(oracle, w, h, 3)
    src, tmp, dst = up(ctx, f0, cw, ch), ctx.plane(cw, ch), ctx.plane(cw, ch)
    ctx.resample_x(src, tmp, ow, h, w)
    ctx.resample_y(tmp, dst, ow, oh, h)
    want = oracle.resample(in_container(f0, cw, ch), w, h, ow, oh)[:oh, :ow]
    assert np.array_equal(dst.download(ow, oh), want)


@pytest.mark.parametrize("w,h,ow,oh", [(37, 20, 100, 70), (33, 17, 34, 18), (50, 35, 100, 70), (64, 64, 64, 64), (2, 3, 257, 130),
                                       (512, 270, 1024, 540), (231, 130, 461, 260), (100, 70, 37, 20), (300, 40, 150, 80),
                                       (1000, 37, 77, 90), (4095, 3, 4096, 5), (3, 4093, 4, 4096), (2047, 9, 2049, 10)])
def test_resample_xy(ctx, oracle, w, h, ow, oh):
    """Both passes in one launch (no temp plane): the bits of the x pass into a temp followed by the y pass, for one plane
    and for two, up-sampling (what the operator routes here) and any other ratio."""
    cw, ch = max(w, ow) + 3, max(h, oh) + 2
    f0, f1, *_ = level_fields(oracle, w, h, 3)
    a, b = up(ctx, f0, cw, ch), up(ctx, f1, cw, ch)
    da, db = ctx.plane(cw, ch).fill_bytes(0x7f), ctx.plane(cw, ch).fill_bytes(0x7f)
    ctx.resample_xy(a, da, w, h, ow, oh, b, db)
    want_a = oracle.resample(in_container(f0, cw, ch), w, h, ow, oh)[:oh, :ow]
    want_b = oracle.resample(in_container(f1, cw, ch), w, h, ow, oh)[:oh, :ow]
    assert np.array_equal(da.download(ow, oh), want_a)
    assert np.array_equal(db.download(ow, oh), want_b)
    got = da.download()
    assert np.all(got[oh:, :].view(np.uint32) == 0x7f7f7f7f) and np.all(got[:, ow:].view(np.uint32) == 0x7f7f7f7f)
    single = ctx.plane(cw, ch).fill_bytes(0x7f)
    ctx.resample_xy(a, single, w, h, ow, oh)
    assert np.array_equal(single.download(ow, oh), want_a)
    tmp, two = ctx.plane(cw, ch), ctx.plane(cw, ch)
    ctx.resample_x(a, tmp, ow, h, w)
    ctx.resample_y(tmp, two, ow, oh, h)
    assert np.array_equal(two.download(ow, oh), want_a)


@pytest.mark.parametrize("w,h,scale,levels", [(8192, 5, 0.5, 11), (4096, 24, 0.5, 7), (1920, 17, 0.5, 7), (1000, 33, 0.45, 5), (257, 40, 0.3, 3),
                                              (640, 12, 0.5, 2), (100, 70, 0.33, 3)])
def test_resample_x_levels(ctx, flow2d, oracle, w, h, scale, levels):
    """The x pass of all pyramid levels in one trip over the frame: every level's segment of the packed plane is
    bit-identical to the single-level x pass (resample_2d.cu:34-75), for both planes of the launch."""
    f0, f1, *_ = level_fields(oracle, w, h, 33)
    widths = [int(np.ceil(np.float32(w) * np.float32(scale) ** np.float32(l))) for l in range(levels, 0, -1)]
    columns, col = [], 0
    for lw in widths:
        columns.append(col)
        col += (lw + 3) // 4 * 4
    assert col <= ctx.plane(w, h).pitch // 4
    a, b = up(ctx, f0, w, h), up(ctx, f1, w, h)
    pa, pb = ctx.plane(w, h).fill_bytes(0x7f), ctx.plane(w, h).fill_bytes(0x7f)
    ctx.resample_x_levels(a, pa, w, h, widths, columns, b, pb)
    ga, gb = pa.download(), pb.download()
    for lw, c in zip(widths, columns):
        assert np.array_equal(ga[:, c:c + lw], oracle.resample_x(f0, lw, h, w)[:, :lw]), lw
        assert np.array_equal(gb[:, c:c + lw], oracle.resample_x(f1, lw, h, w)[:, :lw]), lw
    single = ctx.plane(w, h).fill_bytes(0x7f)  # one plane, one level
    ctx.resample_x_levels(a, single, w, h, widths[:1], [0])
    assert np.array_equal(single.download()[:, :widths[0]], oracle.resample_x(f0, widths[0], h, w)[:, :widths[0]])
    with pytest.raises(flow2d.Flow2DError):  # overlapping segments
        ctx.resample_x_levels(a, pa, w, h, widths[:2], [0, 0])


@pytest.mark.parametrize("w,h,scale,levels", [(4096, 512, 0.5, 7), (1920, 1080, 0.5, 7), (1000, 330, 0.45, 5), (257, 400, 0.3, 3), (640, 120, 0.5, 2),
                                              (100, 70, 0.33, 3), (2048, 2048, 0.5, 10)])
def test_resample_y_levels(ctx, flow2d, oracle, w, h, scale, levels):
    """Round 6: the y passes of all pyramid levels in one launch (flow2d_resample_y_levels) behind the x passes of all levels: every
    level's plane region is bit-identical to the two-pass resample of that level alone (resample_2d.cu:34-118), for both planes."""
    f0, f1, *_ = level_fields(oracle, w, h, 35)
    widths = [int(np.ceil(np.float32(w) * np.float32(scale) ** np.float32(l))) for l in range(levels, 0, -1)]
    heights = [int(np.ceil(np.float32(h) * np.float32(scale) ** np.float32(l))) for l in range(levels, 0, -1)]
    columns, col, rows, row = [], 0, [], 0
    for lw, lh in zip(widths, heights):
        columns.append(col)
        col += (lw + 3) // 4 * 4
        rows.append(row)
        row += lh
    assert col <= ctx.plane(w, h).pitch // 4 and row <= h
    a, b = up(ctx, f0, w, h), up(ctx, f1, w, h)
    pa, pb = ctx.plane(w, h).fill_bytes(0x7f), ctx.plane(w, h).fill_bytes(0x7f)
    ctx.resample_x_levels(a, pa, w, h, widths, columns, b, pb)
    oa, ob = ctx.plane(w, h).fill_bytes(0x7f), ctx.plane(w, h).fill_bytes(0x7f)
    ctx.resample_y_levels(pa, oa, h, widths, heights, columns, rows, pb, ob)
    ga, gb = oa.download(), ob.download()
    for lw, lh, r in zip(widths, heights, rows):
        assert np.array_equal(ga[r:r + lh, :lw], oracle.resample(f0, w, h, lw, lh)[:lh, :lw]), (lw, lh)
        assert np.array_equal(gb[r:r + lh, :lw], oracle.resample(f1, w, h, lw, lh)[:lh, :lw]), (lw, lh)
    assert np.all(ga[row:, :].view(np.uint32) == 0x7f7f7f7f)  # nothing written below the last level
    single = ctx.plane(w, h).fill_bytes(0x7f)  # one plane, one level
    ctx.resample_y_levels(pa, single, h, widths[-1:], heights[-1:], columns[-1:], [0])
    assert np.array_equal(single.download()[:heights[-1], :widths[-1]], oracle.resample(f0, w, h, widths[-1], heights[-1])[:heights[-1], :widths[-1]])
    if levels > 1:
        with pytest.raises(flow2d.Flow2DError):  # overlapping output regions
            ctx.resample_y_levels(pa, oa, h, widths[:2], heights[:2], columns[:2], [0, 0])


@pytest.mark.parametrize("hx,hy", [(1.0, 1.0), (1.25, 1.1), (7.3, 5.5)])
@pytest.mark.parametrize("w,h,cw,ch", SIZES)
def test_registration(ctx, oracle, w, h, cw, ch, hx, hy):
    f0, f1, u, v, *_ = level_fields(oracle, w, h, 4, flow_scale=4.0)
    u[0, 0] = np.nan           # NaN and far out-of-range displacements fall back to frame_0
    v[h - 1, w - 1] = 1e9
    u[h // 2, w // 2] = -1e9
    planes = [up(ctx, a, cw, ch) for a in (f0, f1, u, v)]
    out = ctx.plane(cw, ch)
    ctx.registration(*planes, w, h, hx, hy, out)
    assert np.array_equal(out.download(w, h), oracle.registration(f0, f1, u, v, w, h, hx, hy))


@pytest.mark.parametrize("w,h", [(1, 1), (1, 7), (2, 1), (2, 9), (3, 3), (63, 5), (64, 2), (65, 3), (129, 4)])
def test_registration_narrow_levels(ctx, oracle, w, h):
    """The warp's gathers come in column pairs (x, x + 1) clamped into the level: widths of one, two and three columns, displacements
    that land in the last column and beyond, widths around the wave size."""
    rng = np.random.default_rng(w * 100 + h)
    f0 = rng.normal(0, 1, (h, w)).astype(np.float32)
    f1 = rng.normal(0, 1, (h, w)).astype(np.float32)
    u = rng.uniform(-1.5 * w, 1.5 * w, (h, w)).astype(np.float32)
    v = rng.uniform(-1.5 * h, 1.5 * h, (h, w)).astype(np.float32)
    u[0, w - 1] = 0.0                       # exactly the last column
    u[h - 1, 0] = np.float32(w - 1)         # from the first column onto the last
    v[h - 1, 0] = 0.0
    cw, ch = w + 5, h + 3
    planes = [up(ctx, a, cw, ch, 5.0) for a in (f0, f1, u, v)]
    out = ctx.plane(cw, ch).fill_bytes(0x7f)
    ctx.registration(*planes, w, h, 1.0, 1.0, out)
    assert np.array_equal(out.download(w, h), oracle.registration(f0, f1, u, v, w, h, 1.0, 1.0))


@pytest.mark.parametrize("w,h,ow,oh,hx,hy", [(37, 20, 100, 70, 1.0, 1.0), (33, 17, 34, 18, 2.0, 2.0), (50, 35, 100, 70, 1.25, 1.1),
                                             (512, 270, 1024, 540, 4.0, 4.0), (231, 130, 461, 260, 7.3, 5.5), (2, 3, 257, 130, 1.0, 1.0),
                                             (100, 70, 37, 20, 3.0, 3.5), (2047, 9, 2049, 10, 1.0, 1.0), (64, 64, 64, 64, 1.0, 1.0),
                                             (33, 17, 66, 34, 2.0, 2.0), (1000, 3, 2000, 6, 1.0, 1.0), (1, 1, 2, 2, 1.0, 1.0),
                                             (1024, 1024, 2048, 2048, 2.0, 2.0)])
def test_upsample_registration(ctx, flow2d, oracle, w, h, ow, oh, hx, hy):
    """The previous level's flow resampled to the level's size and frame 1 warped by it in ONE launch: (u, v) are the bits of the
    two-pass resample (resample_2d.cu:34-118), the warped frame the bits of registration_2d.cu:34-73 fed with them -- NaN and far
    out-of-range displacements (frame 0's value) included; nothing outside the level is written.  (Levels exactly twice the previous
    one take a form of their own: one load per plane for a 2 x 2 block of outputs.)"""
    cw, ch = max(w, ow) + 3, max(h, oh) + 2
    _, _, u, v, *_ = level_fields(oracle, w, h, 9, flow_scale=4.0)
    u[0, 0] = np.nan
    v[h - 1, w - 1] = 1e9
    u[h // 2, w // 2] = -1e9
    f0, f1, *_ = level_fields(oracle, ow, oh, 10)
    pu, pv, p0, p1 = up(ctx, u, cw, ch), up(ctx, v, cw, ch), up(ctx, f0, cw, ch), up(ctx, f1, cw, ch)
    ou, ov, out = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(3))
    ctx.upsample_registration(pu, pv, w, h, ou, ov, p0, p1, ow, oh, hx, hy, out)
    want_u = oracle.resample(in_container(u, cw, ch), w, h, ow, oh)[:oh, :ow]
    want_v = oracle.resample(in_container(v, cw, ch), w, h, ow, oh)[:oh, :ow]
    assert np.array_equal(ou.download(ow, oh), want_u, equal_nan=True)
    assert np.array_equal(ov.download(ow, oh), want_v, equal_nan=True)
    want = oracle.registration(f0, f1, np.ascontiguousarray(want_u), np.ascontiguousarray(want_v), ow, oh, hx, hy)
    assert np.array_equal(out.download(ow, oh), want)
    for plane in (ou, ov, out):
        got = plane.download()
        assert np.all(got[oh:, :].view(np.uint32) == 0x7f7f7f7f) and np.all(got[:, ow:].view(np.uint32) == 0x7f7f7f7f)
    # the two launches it replaces
    tu, tv, tout = (ctx.plane(cw, ch) for _ in range(3))
    ctx.resample_xy(pu, tu, w, h, ow, oh, pv, tv)
    ctx.registration(p0, p1, tu, tv, ow, oh, hx, hy, tout)
    assert np.array_equal(out.download(ow, oh), tout.download(ow, oh))
    # the coarsest level's form: no previous flow, zeros stored over the level's region (only) and warped by
    zu, zv, zout = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(3))
    ctx.upsample_registration(None, None, 0, 0, zu, zv, p0, p1, ow, oh, hx, hy, zout)
    zeros = np.zeros((oh, ow), np.float32)
    for plane in (zu, zv):
        got = plane.download()
        assert np.all(got[:oh, :ow].view(np.uint32) == 0)
        assert np.all(got[oh:, :].view(np.uint32) == 0x7f7f7f7f) and np.all(got[:, ow:].view(np.uint32) == 0x7f7f7f7f)
    assert np.array_equal(zout.download(ow, oh).view(np.uint32), oracle.registration(f0, f1, zeros, zeros, ow, oh, hx, hy).view(np.uint32))
    with pytest.raises(flow2d.Flow2DError):  # no previous flow, but a previous size
        ctx.upsample_registration(None, None, w, h, zu, zv, p0, p1, ow, oh, hx, hy, zout)
    with pytest.raises(flow2d.Flow2DError):  # a written plane that is also read
        ctx.upsample_registration(pu, pv, w, h, pu, ov, p0, p1, ow, oh, hx, hy, out)
    with pytest.raises(flow2d.Flow2DError):  # two written planes that are one
        ctx.upsample_registration(pu, pv, w, h, ou, ou, p0, p1, ow, oh, hx, hy, out)


@pytest.mark.parametrize("hx,hy", [(1.0, 1.0), (1.25, 1.1)])
@pytest.mark.parametrize("w,h,cw,ch", SIZES)
def test_phi_ksi_and_sweeps(ctx, flow2d, oracle, w, h, cw, ch, hx, hy):
    f0, f1, u, v, du, dv = level_fields(oracle, w, h, 5)
    d = [up(ctx, a, cw, ch, 3.0) for a in (f0, f1, u, v, du, dv)]
    phi, ksi, tdu, tdv = (ctx.plane(cw, ch) for _ in range(4))
    ctx.compute_phi_ksi(*d, w, h, hx, hy, 0.001, 0.001, phi, ksi)
    ophi, oksi = oracle.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, hx, hy, 0.001, 0.001)
    assert np.array_equal(phi.download(w, h), ophi)
    assert np.array_equal(ksi.download(w, h), oksi)
    for constancy in (flow2d.GREY, flow2d.GRADIENT, flow2d.GRADIENT_UNTILED):
        ctx.solve_sweep(*d, phi, ksi, w, h, hx, hy, 35.0, tdu, tdv, constancy)
        odu, odv = oracle.solve_sweep(f0, f1, u, v, du, dv, ophi, oksi, w, h, hx, hy, 35.0, constancy)
        assert np.array_equal(tdu.download(w, h), odu), "du constancy %d" % constancy
        assert np.array_equal(tdv.download(w, h), odv), "dv constancy %d" % constancy


@pytest.mark.parametrize("w,h,cw,ch,hx,hy", [(4096, 2050, 4096, 2050, 1.0, 1.0), (2777, 3100, 2816, 3104, 1.25, 1.1),
                                              (16000, 530, 16000, 530, 7.3, 5.5), (1000, 300, 1024, 300, 1.0, 1.0)])
def test_sweep_streaming_strips(ctx, flow2d, oracle, w, h, cw, ch, hx, hy):
    """solve_2d and solve_2d_grad at level sizes that take the streaming form (8 Mpixel and more: strips of 64 aligned columns
    walking down the level, windows in registers, lane shifts for the x neighbours with the strip's halo columns from two-lane
    loads, mirrored halo loads instead of border selects): partial last strips, strip heights that do not divide the level,
    levels inside larger containers, spacings that are no powers of two -- bit-identical to the oracle's sweep, two sweeps in
    a row (the second reads what the first wrote), then a red-black SOR iteration in place (its half-sweeps stream too).  (The
    last size stays on the tile form: the same checks.)"""
    f0, f1, u, v, du, dv = level_fields(oracle, w, h, 71)
    d = [up(ctx, a, cw, ch, 3.0) for a in (f0, f1, u, v, du, dv)]
    phi, ksi, tdu, tdv = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(4))
    ctx.compute_phi_ksi(*d, w, h, hx, hy, 0.001, 0.001, phi, ksi)
    ophi, oksi = oracle.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, hx, hy, 0.001, 0.001)
    for constancy in (flow2d.GREY, flow2d.GRADIENT, flow2d.GRADIENT_UNTILED, flow2d.LOG_DERIVATIVES):
        d[4].upload(in_container(du, cw, ch, 3.0)), d[5].upload(in_container(dv, cw, ch, 3.0))
        tdu.fill_bytes(0x7f), tdv.fill_bytes(0x7f)
        ctx.solve_sweep(*d, phi, ksi, w, h, hx, hy, 35.0, tdu, tdv, constancy)
        odu, odv = oracle.solve_sweep(f0, f1, u, v, du, dv, ophi, oksi, w, h, hx, hy, 35.0, constancy)
        if constancy == flow2d.LOG_DERIVATIVES:
            # log(I + 1) comes from the device library here and from the CPU's libm in the oracle (last-place differences);
            # the bit-for-bit check of the streaming log form is against the reference's own kernel (test_gpu_reference.py)
            assert float(np.abs(tdu.download(w, h) - odu).max()) < 1e-5 and float(np.abs(tdv.download(w, h) - odv).max()) < 1e-5
            continue
        assert np.array_equal(tdu.download(w, h), odu) and np.array_equal(tdv.download(w, h), odv), constancy
        # nothing outside the level is written
        full = tdu.download()
        assert np.all(full[h:, :].view(np.uint32) == 0x7f7f7f7f) and np.all(full[:h, w:].view(np.uint32) == 0x7f7f7f7f)
        ctx.solve_sweep(d[0], d[1], d[2], d[3], tdu, tdv, phi, ksi, w, h, hx, hy, 35.0, d[4], d[5], constancy)
        odu2, odv2 = oracle.solve_sweep(f0, f1, u, v, odu, odv, ophi, oksi, w, h, hx, hy, 35.0, constancy)
        assert np.array_equal(d[4].download(w, h), odu2) and np.array_equal(d[5].download(w, h), odv2), constancy
        # opt-in red-black SOR, in place on what the two sweeps left
        ctx.sor_iteration(*d, phi, ksi, w, h, hx, hy, 35.0, 1.6, constancy)
        odu3, odv3 = oracle.sor_iteration(f0, f1, u, v, odu2, odv2, ophi, oksi, w, h, hx, hy, 35.0, 1.6, constancy)
        got = d[4].download()
        assert np.array_equal(got[:h, :w], odu3) and np.array_equal(d[5].download(w, h), odv3), ("sor", constancy)
        assert np.all(got[h:, :] == 3.0) and np.all(got[:, w:] == 3.0)


@pytest.mark.parametrize("window", [3, 5, 7])
@pytest.mark.parametrize("w,h,cw,ch", SIZES)
def test_median(ctx, oracle, w, h, cw, ch, window):
    _, _, u, *_ = level_fields(oracle, w, h, 6)
    u[::3, ::5] = 0.0  # ties
    src, dst = up(ctx, u, cw, ch, 99.0), ctx.plane(cw, ch)
    ctx.median(src, w, h, window, dst)
    assert np.array_equal(dst.download(w, h), oracle.median(u, w, h, window))


@pytest.mark.parametrize("w,h,cw,ch", [(700, 133, 704, 140), (1000, 300, 1024, 300), (61, 200, 64, 200), (12, 12, 16, 16),
                                        (59, 13, 64, 16), (117, 67, 128, 70)])
def test_median7_streaming_strips(ctx, flow2d, oracle, w, h, cw, ch):
    """Window 7 runs a streaming kernel too (round 4): sorted 7-tuples by lane shifts, eight rows in registers, the generated
    pair network for two vertically adjacent medians.  Interior strips (lane neighbours), border strips (mirrored loads),
    odd heights, strip ends inside a pair of rows, many ties, NaNs and -0 (windows redone in the reference's sort order);
    and the add-in-median form (the pyramid's u += du folded into the filter)."""
    _, _, u, v, du, dv = level_fields(oracle, w, h, 17)
    u[::7, ::3] = 0.0
    u[1::5, :] = np.round(u[1::5, :] * 4) / 4  # many ties
    src, dst = up(ctx, u, cw, ch, -5.0), ctx.plane(cw, ch).fill_bytes(0x7f)
    ctx.median(src, w, h, 7, dst)
    assert np.array_equal(dst.download(w, h), oracle.median(u, w, h, 7))
    full = dst.download()
    assert np.all(full[h:, :].view(np.uint32) == 0x7f7f7f7f) and np.all(full[:h, w:].view(np.uint32) == 0x7f7f7f7f)
    u2 = u.copy()
    u2[h // 2, w // 3] = np.nan
    u2[h // 3, :: 11] = -0.0
    u2[0, 0] = np.nan
    src2 = up(ctx, u2, cw, ch, 1.0)
    ctx.median(src2, w, h, 7, dst)
    assert np.array_equal(dst.download(w, h).view(np.uint32), oracle.median(u2, w, h, 7).view(np.uint32))
    # u + du and v + dv through the filter in one launch
    a, b, da, db = (up(ctx, q, cw, ch, 2.0) for q in (u, v, du, dv))
    oa, ob = ctx.plane(cw, ch), ctx.plane(cw, ch)
    ctx.add_median(a, da, w, h, 7, oa, b, db, ob)
    assert np.array_equal(oa.download(w, h), oracle.median(u + du, w, h, 7))
    assert np.array_equal(ob.download(w, h), oracle.median(v + dv, w, h, 7))


@pytest.mark.parametrize("w,h,cw,ch", [(700, 133, 704, 140), (1000, 300, 1024, 300), (61, 200, 64, 200), (8, 8, 8, 8)])
def test_median5_streaming_strips(ctx, oracle, w, h, cw, ch):
    """Window 5 runs the streaming kernel: interior strips (lane neighbours), border strips (mirrored loads),
    odd heights, strip heights from 8 to 64 rows."""
    _, _, u, *_ = level_fields(oracle, w, h, 16)
    u[::7, ::3] = 0.0
    u[1::5, :] = np.round(u[1::5, :] * 4) / 4  # many ties
    src, dst = up(ctx, u, cw, ch, -5.0), ctx.plane(cw, ch)
    ctx.median(src, w, h, 5, dst)
    assert np.array_equal(dst.download(w, h), oracle.median(u, w, h, 5))


@pytest.mark.parametrize("window", [3, 5, 7])
@pytest.mark.parametrize("w,h,cw,ch", [(100, 70, 128, 80), (257, 33, 300, 40), (700, 133, 704, 140), (8, 8, 8, 8), (7, 5, 8, 8)])
def test_median_nan_and_signed_zero(ctx, oracle, w, h, cw, ch, window):
    """Windows holding NaNs or zeros of both signs: the one place where the ORDER of the reference's insertion
    sort (`temp < window[j]`, median_2d.cu:52-63) shows.  The oracle restates that sort (and is held to the
    reference's own kernel on such inputs by tests/golden/ref_kernels_golden.npz); the HIP kernels detect those
    windows and evaluate the same rule: identical bits, NaN payloads and zero signs included."""
    rng = np.random.default_rng(w * 131 + h)
    _, _, u, *_ = level_fields(oracle, w, h, 26)
    u[rng.random((h, w)) < 0.15] = 0.0
    u[rng.random((h, w)) < 0.10] = -0.0
    u[rng.random((h, w)) < 0.03] = np.nan
    u[0, 0] = np.nan
    u[h - 1, w - 1] = -0.0
    u[h // 2, : min(w, 6)] = np.nan           # a run of NaNs: windows cut into several runs
    src, dst = up(ctx, u, cw, ch, 99.0), ctx.plane(cw, ch)
    ctx.median(src, w, h, window, dst)
    got, want = dst.download(w, h), oracle.median(u, w, h, window)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # and a plane without any: untouched by the detection
    clean = np.abs(np.nan_to_num(u)) + 1.0
    src2 = up(ctx, clean.astype(np.float32), cw, ch, 99.0)
    ctx.median(src2, w, h, window, dst)
    assert np.array_equal(dst.download(w, h), oracle.median(clean.astype(np.float32), w, h, window))


@pytest.mark.parametrize("w,h,cw,ch", [(100, 70, 128, 80), (257, 33, 300, 40), (16, 8, 16, 8), (700, 133, 704, 140)])
def test_two_plane_launches(ctx, flow2d, oracle, w, h, cw, ch):
    """add / median / resample on two planes per launch: each plane gets what the single-plane entry gives."""
    a, b, c, d, *_ = level_fields(oracle, w, h, 21)
    pa, pb, pc, pd = (up(ctx, x, cw, ch, 3.0) for x in (a, b, c, d))
    ctx.add_pair(pa, pb, pc, pd, w, h)
    assert np.array_equal(pa.download(w, h), oracle.add(a, b, w, h)) and np.array_equal(pc.download(w, h), oracle.add(c, d, w, h))
    for window in (3, 5, 7):
        sa, sb, da, db = up(ctx, a, cw, ch), up(ctx, c, cw, ch), ctx.plane(cw, ch), ctx.plane(cw, ch)
        ctx.median_pair(sa, sb, w, h, window, da, db)
        assert np.array_equal(da.download(w, h), oracle.median(a, w, h, window))
        assert np.array_equal(db.download(w, h), oracle.median(c, w, h, window))
    for ow, oh in ((w // 2, h // 2), (max(2, w // 9), max(2, h // 5)), (min(cw, w + 20), min(ch, h + 7))):
        sa, sb = up(ctx, a, cw, ch), up(ctx, c, cw, ch)
        ta, tb, da, db = (ctx.plane(cw, ch) for _ in range(4))
        ctx.resample_x_pair(sa, ta, sb, tb, ow, h, w)
        ctx.resample_y_pair(ta, da, tb, db, ow, oh, h)
        assert np.array_equal(da.download(ow, oh), oracle.resample(in_container(a, cw, ch), w, h, ow, oh)[:oh, :ow])
        assert np.array_equal(db.download(ow, oh), oracle.resample(in_container(c, cw, ch), w, h, ow, oh)[:oh, :ow])
    with pytest.raises(flow2d.Flow2DError):
        ctx.median_pair(pa, pb, w, h, 5, pc, pc)  # the two outputs must differ


@pytest.mark.parametrize("window", [3, 5, 7])
@pytest.mark.parametrize("w,h,cw,ch", [(100, 70, 128, 80), (700, 133, 704, 140), (1000, 300, 1024, 300), (8, 8, 8, 8), (7, 5, 8, 8)])
def test_add_median(ctx, flow2d, oracle, w, h, cw, ch, window):
    """`u += du` and the median of u in one launch (the filter reads u + du as it goes): the bits of add_2d followed by
    median_2d, sums that are NaN or -0 included (u = -0 and du = -0 is the only way to a -0 sum), for one plane and for
    two; the inputs are left as they were."""
    rng = np.random.default_rng(w * 7 + h + window)
    _, _, u, v, *_ = level_fields(oracle, w, h, 36)
    du = (rng.normal(0, 0.3, (h, w))).astype(np.float32)
    dv = (rng.normal(0, 0.3, (h, w))).astype(np.float32)
    du[::3, ::4] = -u[::3, ::4]            # sums of exactly +0: ties
    u[1::4, 1::5] = -0.0
    du[1::4, 1::5] = -0.0                  # sums of -0
    du[rng.random((h, w)) < 0.02] = np.nan
    du[h // 2, : min(w, 6)] = np.nan       # a run of NaN sums (propagated NaNs: the payload is the operand's on both sides)
    pu, pdu, pv, pdv = (up(ctx, a, cw, ch, 99.0) for a in (u, du, v, dv))
    ou, ov = ctx.plane(cw, ch).fill_bytes(0x7f), ctx.plane(cw, ch).fill_bytes(0x7f)
    ctx.add_median(pu, pdu, w, h, window, ou, pv, pdv, ov)
    want_u = oracle.median(oracle.add(u, du, w, h), w, h, window)
    want_v = oracle.median(oracle.add(v, dv, w, h), w, h, window)
    assert np.array_equal(ou.download(w, h).view(np.uint32), want_u.view(np.uint32))
    assert np.array_equal(ov.download(w, h).view(np.uint32), want_v.view(np.uint32))
    assert np.array_equal(pu.download(w, h).view(np.uint32), u.view(np.uint32))  # not modified
    single = ctx.plane(cw, ch).fill_bytes(0x7f)
    ctx.add_median(pv, pdv, w, h, window, single)
    assert np.array_equal(single.download(w, h).view(np.uint32), want_v.view(np.uint32))
    got = single.download()
    assert np.all(got[h:, :].view(np.uint32) == 0x7f7f7f7f) and np.all(got[:, w:].view(np.uint32) == 0x7f7f7f7f)
    with pytest.raises(flow2d.Flow2DError):
        ctx.add_median(pu, pdu, w, h, window, pdu)  # the addend cannot be the output


def test_median_rejects_bad_window(ctx, flow2d, oracle):
    src, dst = ctx.plane(32, 32), ctx.plane(32, 32)
    for bad in (0, 1, 2, 4, 9):
        with pytest.raises(flow2d.Flow2DError) as e:
            ctx.median(src, 32, 32, bad, dst)
        assert e.value.status == 5
    with pytest.raises(flow2d.Flow2DError) as e:
        ctx.median(src, 32, 32, 5, src)  # in == out
    assert e.value.status == 1


@pytest.mark.parametrize("algorithm", [1, 2, 4, 0])
@pytest.mark.parametrize("constancy", [0, 1, 2])  # 2 = gradient term over true neighbours (not in the reference)
@pytest.mark.parametrize("outer,inner", [(2, 3), (3, 2), (1, 5), (2, 1), (1, 4)])
@pytest.mark.parametrize("w,h,cw,ch", SIZES[:4] + [(300, 150, 320, 160), (52, 64, 64, 64), (53, 65, 64, 80), (640, 520, 640, 520),
                                     (401, 333, 416, 340)])
def test_solve_level(ctx, oracle, w, h, cw, ch, outer, inner, constancy, algorithm):
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 7)
    hx, hy = np.float32(cw / w), np.float32(ch / h)
    d = [up(ctx, a, cw, ch) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(6))
    rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, 3.5, 0.001, 0.001, outer, inner,
                               constancy, algorithm)
    odu, odv, ophi, oksi = oracle.solve_level(f0, f1, u, v, w, h, hx, hy, 3.5, 0.001, 0.001, outer, inner, constancy)
    assert np.array_equal(rdu.download(w, h), odu)
    assert np.array_equal(rdv.download(w, h), odv)
    # which pair holds the result: per-sweep = the reference's swap parity (cuda_operation_solve_2d.cpp:288-289),
    # fused = one swap per outer iteration; either way the library reports it
    single = algorithm == 0 and w <= 64 and h <= 32 and inner < 2  # AUTO: tiles whenever there are sweeps to fuse
    fused = algorithm in (2, 4) or (algorithm == 0 and not single and inner >= 2)  # one launch per outer iteration
    launches = 0 if single else (outer if fused else outer * inner)
    assert (rdu is tdu) == (launches % 2 == 1)


@pytest.mark.parametrize("algorithm", [2, 0])
@pytest.mark.parametrize("constancy", [0, 1, 2])
@pytest.mark.parametrize("outer,inner", [(1, 6), (2, 7), (3, 8), (1, 10), (4, 11), (1, 16)])
@pytest.mark.parametrize("w,h,cw,ch", [(100, 70, 128, 80), (640, 520, 640, 520)])
def test_solve_level_fused_more_than_five_sweeps(ctx, oracle, w, h, cw, ch, outer, inner, constancy, algorithm):
    """More sweeps per outer iteration than one fused launch holds: the launches of one outer iteration share
    their coefficients' source and hand the increment on; the result can end in any of the three plane pairs."""
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 17)
    hx, hy = np.float32(cw / w), np.float32(ch / h)
    d = [up(ctx, a, cw, ch) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(6))
    rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, 3.5, 0.001, 0.001, outer, inner,
                               constancy, algorithm)
    odu, odv, _, _ = oracle.solve_level(f0, f1, u, v, w, h, hx, hy, 3.5, 0.001, 0.001, outer, inner, constancy)
    assert np.array_equal(rdu.download(w, h), odu)
    assert np.array_equal(rdv.download(w, h), odv)


@pytest.mark.parametrize("constancy", [0, 1, 2])  # 2 = gradient term over true neighbours (not in the reference)
@pytest.mark.parametrize("outer,inner", [(2, 3), (3, 5), (1, 7), (2, 0), (0, 3)])
@pytest.mark.parametrize("w,h", [(5, 4), (16, 8), (64, 16), (33, 17), (64, 32), (40, 33), (64, 64), (52, 61), (2, 2)])
def test_solve_level_single_workgroup(ctx, flow2d, oracle, w, h, outer, inner, constancy):
    """The whole-level kernel (levels up to 64 x 64, any iteration counts) against the oracle."""
    cw, ch = 64, 64
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 9)
    hx, hy = np.float32(cw / w), np.float32(ch / h)
    d = [up(ctx, a, cw, ch) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(6))
    rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, 3.5, 0.001, 0.001, outer, inner,
                               constancy, flow2d.SOLVER_SINGLE_WORKGROUP)
    odu, odv, _, _ = oracle.solve_level(f0, f1, u, v, w, h, hx, hy, 3.5, 0.001, 0.001, outer, inner, constancy)
    assert rdu is du
    assert np.array_equal(rdu.download(w, h), odu) and np.array_equal(rdv.download(w, h), odv)


def test_solve_level_single_workgroup_rejects_large_levels(ctx, flow2d, oracle):
    w, h = 65, 40
    planes = [ctx.plane(w, h).fill_bytes(0) for _ in range(10)]
    with pytest.raises(flow2d.Flow2DError) as e:
        ctx.solve_level(*planes, w, h, 1.0, 1.0, 3.5, 0.001, 0.001, 1, 1, 0, flow2d.SOLVER_SINGLE_WORKGROUP)
    assert e.value.status == 5


@pytest.mark.parametrize("algorithm", [1, 2, 4, 0])
@pytest.mark.parametrize("case", ["zero frames", "constant frames", "alpha 0", "epsilon 0", "huge values"])
def test_solve_level_degenerate_inputs(ctx, oracle, case, algorithm):
    """Flat images, vanishing regularisation or robustifier, overflowing intermediates: zeros, infinities and NaNs
    come out where the oracle's IEEE arithmetic puts them."""
    w, h, cw, ch = 150, 90, 160, 96
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 23)
    alpha, eps = 3.5, 0.001
    if case == "zero frames":
        f0, f1 = np.zeros_like(f0), np.zeros_like(f1)
    elif case == "constant frames":
        f0, f1 = np.full_like(f0, 17.0), np.full_like(f1, 17.0)
    elif case == "alpha 0":
        alpha = 0.0
    elif case == "epsilon 0":
        eps = 0.0
        u, v = np.zeros_like(u), np.zeros_like(v)   # zero gradients: phi = 1 / (2 sqrt(0)) = inf
    else:
        f0, f1 = f0 * np.float32(1e18), f1 * np.float32(1e18)
    hx, hy = np.float32(cw / w), np.float32(ch / h)
    d = [up(ctx, a, cw, ch) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(6))
    rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, alpha, eps, eps, 2, 3, 0, algorithm)
    odu, odv, _, _ = oracle.solve_level(f0, f1, u, v, w, h, hx, hy, alpha, eps, eps, 2, 3, 0)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(rdu.download(w, h), odu, equal_nan=True)
        assert np.array_equal(rdv.download(w, h), odv, equal_nan=True)


def test_solve_level_fused_without_sweeps(ctx, flow2d, oracle):
    """FUSED has nothing to fuse when an outer iteration holds no sweep: refused; AUTO runs the per-sweep form."""
    w, h = 64, 48
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 8)
    d = [up(ctx, a, w, h) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h) for _ in range(6))
    with pytest.raises(flow2d.Flow2DError) as e:
        ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 3.5, 0.001, 0.001, 2, 0, 0, 2)
    assert e.value.status == 5
    rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 3.5, 0.001, 0.001, 2, 0, 0, 0)
    odu, odv, _, _ = oracle.solve_level(f0, f1, u, v, w, h, 1.0, 1.0, 3.5, 0.001, 0.001, 2, 0, 0)
    assert np.array_equal(rdu.download(w, h), odu) and np.array_equal(rdv.download(w, h), odv)


# ---- opt-in red-black SOR (no reference counterpart; checked against its own oracle restatement) ---------------
@pytest.mark.parametrize("constancy", [0, 1, 2])  # 2 = gradient term over true neighbours (not in the reference)
@pytest.mark.parametrize("omega", [1.0, 1.5])
@pytest.mark.parametrize("w,h,cw,ch", SIZES[:4])
def test_sor_iteration_and_level(ctx, oracle, w, h, cw, ch, omega, constancy):
    f0, f1, u, v, du, dv = level_fields(oracle, w, h, 31)
    hx, hy = np.float32(1.25), np.float32(1.1)
    d = [up(ctx, a, cw, ch, 3.0) for a in (f0, f1, u, v, du, dv)]
    phi, ksi, tdu, tdv = (ctx.plane(cw, ch) for _ in range(4))
    ctx.compute_phi_ksi(*d, w, h, hx, hy, 0.001, 0.001, phi, ksi)
    ophi, oksi = oracle.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, hx, hy, 0.001, 0.001)
    ctx.sor_iteration(*d, phi, ksi, w, h, hx, hy, 35.0, omega, constancy)
    odu, odv = oracle.sor_iteration(f0, f1, u, v, du, dv, ophi, oksi, w, h, hx, hy, 35.0, omega, constancy)
    got = d[4].download()
    assert np.array_equal(got[:h, :w], odu) and np.array_equal(d[5].download(w, h), odv)
    assert np.all(got[h:, :] == 3.0) and np.all(got[:, w:] == 3.0)  # in place, level rectangle only
    # the level loop with SOR iterations: as half-sweep launches (result stays in du / dv), and temporally blocked in the
    # strip kernel (AUTO and FUSED: two iterations per launch, three iterations = launches of 2 + 1) -- the same bits
    odu, odv = oracle.solve_level_sor(f0, f1, u, v, w, h, hx, hy, 3.5, 0.001, 0.001, 2, 3, omega, constancy)
    for algorithm in (1, 0, 2):
        rdu, rdv = ctx.solve_level(*d[:4], d[4], d[5], phi, ksi, tdu, tdv, w, h, hx, hy, 3.5, 0.001, 0.001, 2, 3, constancy,
                                   algorithm, sor_omega=omega)
        if algorithm == 1:
            assert rdu is d[4]
        assert np.array_equal(rdu.download(w, h), odu) and np.array_equal(rdv.download(w, h), odv), algorithm


def test_sor_rejects_bad_omega_and_kernels_without_half_sweeps(ctx, flow2d):
    w, h = 64, 48
    planes = [ctx.plane(w, h).fill_bytes(0) for _ in range(10)]
    for omega in (2.0, -0.5):
        with pytest.raises(flow2d.Flow2DError) as e:
            ctx.solve_level(*planes, w, h, 1.0, 1.0, 3.5, 0.001, 0.001, 1, 1, 0, 0, sor_omega=omega)
        assert e.value.status == 1
    with pytest.raises(flow2d.Flow2DError) as e:  # the single-workgroup kernel is Jacobi only
        ctx.solve_level(*planes, w, h, 1.0, 1.0, 3.5, 0.001, 0.001, 1, 1, 0, 3, sor_omega=1.2)
    assert e.value.status == 5
    with pytest.raises(flow2d.Flow2DError) as e:  # the tiles hold two iterations (four half-sweep stages) at most
        ctx.solve_level(*planes, w, h, 1.0, 1.0, 3.5, 0.001, 0.001, 1, 3, 0, 4, sor_omega=1.2)
    assert e.value.status == 5
    with pytest.raises(flow2d.Flow2DError) as e:  # solve_2d_log has no red-black form
        ctx.solve_level(*planes, w, h, 1.0, 1.0, 3.5, 0.001, 0.001, 1, 1, 3, 0, sor_omega=1.2)
    assert e.value.status == 5


@pytest.mark.parametrize("constancy", [0, 1, 2])
@pytest.mark.parametrize("w,h,iterations,omega", [(330, 250, 1, 1.9), (330, 250, 2, 1.0), (200, 136, 5, 1.5), (1100, 90, 4, 1.7),
                                                  (64, 700, 3, 0.6)])
def test_sor_in_the_strip_kernel(ctx, oracle, w, h, iterations, omega, constancy):
    """The temporally blocked red-black SOR (round 5): half-sweeps as the stages of the fused strip kernel -- several strips
    per column and per row, borders, one / two / several launches per outer iteration (five iterations = 2 + 2 + 1), power-of-two
    and other grid spacings -- bit for bit against the oracle's in-place restatement."""
    if constancy == 1:
        w, h = (w + 15) // 16 * 16, (h + 7) // 8 * 8  # the reference's tile rule is defined on multiples of 16 x 8
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 77)
    d = [up(ctx, a, w, h) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h) for _ in range(6))
    for hx, hy in ((1.0, 1.0), (np.float32(1.25), np.float32(1.6))):
        rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, 35.0, 0.001, 0.001, 3, iterations, constancy,
                                   2, sor_omega=omega)
        odu, odv = oracle.solve_level_sor(f0, f1, u, v, w, h, hx, hy, 35.0, 0.001, 0.001, 3, iterations, omega, constancy)
        assert np.array_equal(rdu.download(w, h), odu) and np.array_equal(rdv.download(w, h), odv), (hx, hy)
    assert ctx.fused_fallbacks() == 0


@pytest.mark.parametrize("constancy", [0, 1, 2])
@pytest.mark.parametrize("w,h,iterations,omega", [(96, 64, 1, 1.9), (150, 90, 2, 1.2), (330, 250, 2, 1.7), (512, 384, 1, 0.8), (40, 33, 2, 1.5)])
def test_sor_in_the_lds_tiles(ctx, oracle, w, h, iterations, omega, constancy):
    """Red-black SOR in the tiled kernel (round 5: half-sweeps as its stages, one or two iterations per launch; 8 x 8, 16 x 16 and
    32 x 32 tiles), explicitly and as AUTO's choice for these level sizes: the oracle's bits."""
    if constancy == 1:
        w, h = (w + 15) // 16 * 16, (h + 7) // 8 * 8
    f0, f1, u, v, _, _ = level_fields(oracle, w, h, 79)
    d = [up(ctx, a, w, h) for a in (f0, f1, u, v)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, h) for _ in range(6))
    hx, hy = np.float32(1.25), np.float32(1.6)
    odu, odv = oracle.solve_level_sor(f0, f1, u, v, w, h, hx, hy, 35.0, 0.001, 0.001, 3, iterations, omega, constancy)
    for algorithm in (4, 0):
        rdu, rdv = ctx.solve_level(*d, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, 35.0, 0.001, 0.001, 3, iterations, constancy,
                                   algorithm, sor_omega=omega)
        assert np.array_equal(rdu.download(w, h), odu) and np.array_equal(rdv.download(w, h), odv), algorithm


@pytest.mark.parametrize("w,h,cw,ch", [(100, 70, 128, 80), (257, 33, 300, 40), (7, 5, 8, 8), (1024, 300, 1024, 300)])
def test_copy_planes(ctx, flow2d, w, h, cw, ch):
    """flow2d_copy_planes: several independent planes by one launch, only the w x h corner of each container."""
    rng = np.random.default_rng(3)
    n = 5
    data = [rng.normal(0, 1, (ch, cw)).astype(np.float32) for _ in range(n)]
    srcs = [ctx.plane(cw, ch, a) for a in data]
    dsts = [ctx.plane(cw, ch).fill_bytes(0x7f) for _ in range(n)]
    ctx.copy_planes(srcs, dsts, w, h)
    poison = np.frombuffer(b"\x7f\x7f\x7f\x7f", np.float32)[0]
    for a, d in zip(data, dsts):
        got = d.download()
        assert np.array_equal(got[:h, :w], a[:h, :w])
        assert (got[h:, :] == poison).all() and (got[:, w:] == poison).all()  # nothing outside the corner is touched
    with pytest.raises(flow2d.Flow2DError):
        ctx.copy_planes([srcs[0]], [srcs[0]], w, h)  # in == out
