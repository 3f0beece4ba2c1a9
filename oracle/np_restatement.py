"""Second, independent restatement of the hot path in vectorised numpy float32.

TEST INFRASTRUCTURE ONLY (same rules as flow2d_oracle.c).  Its one job is to cross-check the
C oracle: the two were written separately from the reference's formulas and must agree bit
for bit (tests/test_oracle.py).  numpy float32 arithmetic never fuses multiply-adds, so the
operation order below is what is evaluated.

Arrays are tight (h, w) float32 images.  Citations are into the reference repository.
"""
import math

import numpy as np

F = np.float32


def max_warp_level(width, height, scale):
    """optical_flow_base_2d.cpp:36-59"""
    rw = rh = 1
    level = 1
    while F(scale) < F(1.0):
        s = F(np.power(F(scale), F(level)))
        rw = int(np.ceil(F(width) * s))
        rh = int(np.ceil(F(height) * s))
        if rw < 4 or rh < 4:
            break
        level += 1
    if rw == 1 or rh == 1:
        level -= 1
    return level


def level_geometry(width, height, scale, level):
    """optical_flow_2d.cpp:268-272"""
    s = F(np.power(F(scale), F(level)))
    cw = int(np.ceil(F(width) * s))
    ch = int(np.ceil(F(height) * s))
    return cw, ch, F(width) / F(cw), F(height) / F(ch)


def gaussian_taps(sigma):
    """cuda_operation_convolution_2d.cpp:83-112"""
    sigma = F(sigma)
    r = int(F(3) * sigma / F(1.0))
    taps = np.zeros(2 * r + 1, F)
    for i in range(-r, r + 1):
        num = -(F(i * i) * F(1.0) * F(1.0))
        arg = float(num) / (2.0 * float(sigma) * float(sigma))
        amp = 1.0 / (float(sigma) * math.sqrt(2.0 * 3.1415926))
        taps[i + r] = F(amp * math.exp(arg))
    s = F(0.0)
    for t in taps:
        s = F(s + t)
    return (taps / s).astype(F), r


def convolution(img, sigma):
    """convolution_2d.cu:74-261: zero padding, accumulate j=-r..r, rows then columns."""
    taps, r = gaussian_taps(sigma)
    h, w = img.shape
    pad = np.zeros((h, w + 2 * r), F)
    pad[:, r:r + w] = img
    tmp = np.zeros((h, w), F)
    for j in range(-r, r + 1):
        tmp = tmp + taps[r - j] * pad[:, r + j:r + j + w]
    pad = np.zeros((h + 2 * r, w), F)
    pad[r:r + h, :] = tmp
    out = np.zeros((h, w), F)
    for j in range(-r, r + 1):
        out = out + taps[r - j] * pad[r + j:r + j + h, :]
    return out


def _resample_axis0(a, out_n):
    """resample_2d.cu:34-118 along axis 0 of a (in_n, m) array."""
    in_n = a.shape[0]
    delta = F(in_n) / F(out_n)
    norm = F(out_n) / F(in_n)
    out = np.zeros((out_n, a.shape[1]), F)
    for g in range(out_n):
        left_f = F(g) * delta
        right_f = F(g + 1) * delta
        left_i = int(np.floor(left_f))
        right_i = min(in_n, int(np.ceil(right_f)))
        cells = right_i - left_i
        value = np.zeros(a.shape[1], F)
        for j in range(cells):
            frac = F(1.0)
            if j == 0:
                frac = F(left_i + 1) - left_f
            if j == cells - 1:
                frac = right_f - F(left_i + j)
            if cells == 1:
                frac = delta
            value = value + a[left_i + j] * frac
        out[g] = value * norm
    return out


def resample(img, out_w, out_h):
    """cuda_operation_resample_2d.cpp:99-105: x pass then y pass."""
    tmp = np.ascontiguousarray(_resample_axis0(np.ascontiguousarray(img.T), out_w).T)
    return _resample_axis0(tmp, out_h)


def registration(f0, f1, u, v, hx, hy):
    """registration_2d.cu:34-73"""
    h, w = f0.shape
    yy, xx = np.mgrid[0:h, 0:w]
    x_f = xx.astype(F) + (u * (F(1.0) / F(hx)))
    y_f = yy.astype(F) + (v * (F(1.0) / F(hy)))
    with np.errstate(invalid="ignore"):
        bad = (x_f < 0) | (x_f > F(w - 1)) | (y_f < 0) | (y_f > F(h - 1)) | np.isnan(x_f) | np.isnan(y_f)
    xs = np.where(bad, F(0), x_f)
    ys = np.where(bad, F(0), y_f)
    x = np.floor(xs).astype(np.int64)
    y = np.floor(ys).astype(np.int64)
    dx = xs - x.astype(F)
    dy = ys - y.astype(F)
    x1 = np.minimum(w - 1, x + 1)
    y1 = np.minimum(h - 1, y + 1)
    one = F(1.0)
    val = ((one - dx) * (one - dy) * f1[y, x] + dx * (one - dy) * f1[y, x1]
           + (one - dx) * dy * f1[y1, x] + dx * dy * f1[y1, x1])
    return np.where(bad, f0, val).astype(F)


def _halo(a):
    """reflect-without-repeat 1-px halo (solve_2d.cu:75-133)."""
    return np.pad(a, 1, mode="reflect")


def _derivs(f0, f1, hx, hy):
    p0, p1 = _halo(f0), _halo(f1)
    fx = (p0[1:-1, 2:] - p0[1:-1, :-2] + p1[1:-1, 2:] - p1[1:-1, :-2]) / (F(4.0) * F(hx))
    fy = (p0[2:, 1:-1] - p0[:-2, 1:-1] + p1[2:, 1:-1] - p1[:-2, 1:-1]) / (F(4.0) * F(hy))
    ft = f1 - f0
    return fx, fy, ft


def compute_phi_ksi(f0, f1, u, v, du, dv, hx, hy, e_smooth, e_data):
    """solve_2d.cu:43-198"""
    hx, hy, es, ed = F(hx), F(hy), F(e_smooth), F(e_data)
    pu, pv, pdu, pdv = _halo(u), _halo(v), _halo(du), _halo(dv)
    dux = (pu[1:-1, 2:] - pu[1:-1, :-2] + pdu[1:-1, 2:] - pdu[1:-1, :-2]) / (F(2.0) * hx)
    duy = (pu[2:, 1:-1] - pu[:-2, 1:-1] + pdu[2:, 1:-1] - pdu[:-2, 1:-1]) / (F(2.0) * hy)
    dvx = (pv[1:-1, 2:] - pv[1:-1, :-2] + pdv[1:-1, 2:] - pdv[1:-1, :-2]) / (F(2.0) * hx)
    dvy = (pv[2:, 1:-1] - pv[:-2, 1:-1] + pdv[2:, 1:-1] - pdv[:-2, 1:-1]) / (F(2.0) * hy)
    phi = F(1.0) / (F(2.0) * np.sqrt(dux * dux + duy * duy + dvx * dvx + dvy * dvy + es * es))
    fx, fy, ft = _derivs(f0, f1, hx, hy)
    J11, J22, J33, J12, J13, J23 = fx * fx, fy * fy, ft * ft, fx * fy, fx * ft, fy * ft
    s = (J11 * du + J12 * dv + J13) * du + (J12 * du + J22 * dv + J23) * dv + (J13 * du + J23 * dv + J33)
    s = (s > 0).astype(F) * s
    ksi = F(1.0) / (F(2.0) * np.sqrt(s + ed * ed))
    return phi.astype(F), ksi.astype(F)


def _tensor_grey(f0, f1, hx, hy):
    fx, fy, ft = _derivs(f0, f1, hx, hy)
    return fx * fx, fy * fy, fx * fy, fx * ft, fy * ft


def _tensor_grad(f0, f1, hx, hy, tile=(16, 8)):
    """solve_2d.cu:795-884: second derivatives inside 16x8 blocks with edge replication.
    tile=None: true neighbours with the reflect rule at the image border (constancy 2, not a reference mode)."""
    fx, fy, ft = _derivs(f0, f1, hx, hy)
    h, w = f0.shape
    hx_1 = F(1.0 / (2.0 * float(hx)))
    hy_1 = F(1.0 / (2.0 * float(hy)))
    xs = np.arange(w)
    ys = np.arange(h)
    if tile is None:
        reflect = lambda i, n: np.where(np.abs(i) >= n, 2 * n - np.abs(i) - 2, np.abs(i))
        xa, xb, ya, yb = reflect(xs - 1, w), reflect(xs + 1, w), reflect(ys - 1, h), reflect(ys + 1, h)
    else:
        xa = np.where(xs % tile[0] == 0, xs, xs - 1)
        xb = np.where((xs % tile[0] == tile[0] - 1) | (xs == w - 1), xs, xs + 1)
        ya = np.where(ys % tile[1] == 0, ys, ys - 1)
        yb = np.where((ys % tile[1] == tile[1] - 1) | (ys == h - 1), ys, ys + 1)
    fxx = (fx[:, xb] - fx[:, xa]) * hx_1
    fxy = (fx[yb, :] - fx[ya, :]) * hy_1
    fyy = (fy[yb, :] - fy[ya, :]) * hy_1
    fxt = (ft[:, xb] - ft[:, xa]) * hx_1
    fyt = (ft[yb, :] - ft[ya, :]) * hy_1
    return (fxx * fxx + fxy * fxy, fxy * fxy + fyy * fyy, fxx * fxy + fxy * fyy,
            fxx * fxt + fxy * fyt, fxy * fxt + fyy * fyt)


def solve_sweep(f0, f1, u, v, du, dv, phi, ksi, hx, hy, alpha, gradient=False):
    """solve_2d.cu:200-377 (Grey) / :683-952 (Gradient): one Jacobi sweep."""
    hx, hy, alpha = F(hx), F(hy), F(alpha)
    h, w = f0.shape
    if gradient == 2:  # untiled
        J11, J22, J12, J13, J23 = _tensor_grad(f0, f1, hx, hy, tile=None)
    else:
        J11, J22, J12, J13, J23 = (_tensor_grad if gradient else _tensor_grey)(f0, f1, hx, hy)
    hx_2 = alpha / (hx * hx)
    hy_2 = alpha / (hy * hy)
    xs = np.arange(w)[None, :]
    ys = np.arange(h)[:, None]
    xp = (xs < w - 1).astype(F) * hx_2 * np.ones((h, 1), F)
    xm = (xs > 0).astype(F) * hx_2 * np.ones((h, 1), F)
    yp = (ys < h - 1).astype(F) * hy_2 * np.ones((1, w), F)
    ym = (ys > 0).astype(F) * hy_2 * np.ones((1, w), F)
    pp = _halo(phi)
    two = F(2.0)
    phi_xp = (pp[1:-1, 2:] + phi) / two
    phi_xm = (pp[1:-1, :-2] + phi) / two
    phi_yp = (pp[2:, 1:-1] + phi) / two
    phi_ym = (pp[:-2, 1:-1] + phi) / two
    sumH = xp * phi_xp + xm * phi_xm + yp * phi_yp + ym * phi_ym
    pu, pv, pdu, pdv = _halo(u), _halo(v), _halo(du), _halo(dv)
    sumU = (phi_xp * xp * (pu[1:-1, 2:] + pdu[1:-1, 2:] - u) + phi_xm * xm * (pu[1:-1, :-2] + pdu[1:-1, :-2] - u)
            + phi_yp * yp * (pu[2:, 1:-1] + pdu[2:, 1:-1] - u) + phi_ym * ym * (pu[:-2, 1:-1] + pdu[:-2, 1:-1] - u))
    sumV = (phi_xp * xp * (pv[1:-1, 2:] + pdv[1:-1, 2:] - v) + phi_xm * xm * (pv[1:-1, :-2] + pdv[1:-1, :-2] - v)
            + phi_yp * yp * (pv[2:, 1:-1] + pdv[2:, 1:-1] - v) + phi_ym * ym * (pv[:-2, 1:-1] + pdv[:-2, 1:-1] - v))
    r_du = (ksi * (-J13 - J12 * dv) + sumU) / (ksi * J11 + sumH)
    r_dv = (ksi * (-J23 - J12 * r_du) + sumV) / (ksi * J22 + sumH)
    return r_du.astype(F), r_dv.astype(F)


def median(img, radius):
    """median_2d.cu:87-299: r x r window, mirror borders, element r*r/2 of the sorted window."""
    r2 = radius // 2
    h, w = img.shape
    p = np.pad(img, r2, mode="reflect")
    stack = np.stack([p[dy:dy + h, dx:dx + w] for dy in range(radius) for dx in range(radius)], axis=0)
    stack.sort(axis=0, kind="stable")
    return np.ascontiguousarray(stack[(radius * radius) // 2])


def compute_flow(frame_0, frame_1, levels, scale, outer, inner, alpha, e_smooth, e_data, median_radius, sigma,
                 gradient=False):
    """optical_flow_2d.cpp:142-569 + cuda_operation_solve_2d.cpp:229-300 (tight images, no containers)."""
    f0 = np.asarray(frame_0, F)
    f1 = np.asarray(frame_1, F)
    H, W = f0.shape
    level = min(levels, max_warp_level(W, H, scale)) - 1
    if sigma > 0:
        f0, f1 = convolution(f0, sigma), convolution(f1, sigma)
    u = v = None
    while level >= 0:
        cw, ch, hx, hy = level_geometry(W, H, scale, level)
        if level == 0:
            g0, g1 = f0, f1
        else:
            g0, g1 = resample(f0, cw, ch), resample(f1, cw, ch)
        if u is None:
            u, v = np.zeros((ch, cw), F), np.zeros((ch, cw), F)
        else:
            u, v = resample(u, cw, ch), resample(v, cw, ch)
        g1 = registration(g0, g1, u, v, hx, hy)
        du, dv = np.zeros((ch, cw), F), np.zeros((ch, cw), F)
        for _ in range(outer):
            phi, ksi = compute_phi_ksi(g0, g1, u, v, du, dv, hx, hy, e_smooth, e_data)
            for _ in range(inner):
                du, dv = solve_sweep(g0, g1, u, v, du, dv, phi, ksi, hx, hy, alpha, gradient)
        u, v = u + du, v + dv
        if median_radius != 1:
            r = median_radius - 1 if median_radius % 2 == 0 else median_radius
            u, v = median(u, r), median(v, r)
        level -= 1
    return u, v
