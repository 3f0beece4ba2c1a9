"""ctypes front end of the CPU oracle (oracle/flow2d_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  Parity status: see the header of
flow2d_oracle.c (pinned bit for bit against the reference's own kernels run on an MI355X and
its own host code run in the build container: tests/golden/ref_*_golden.npz).

Planes are C-contiguous float32 arrays of shape (container_h, pitch); a level occupies the
top-left w x h corner, like the reference's pitched containers.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libflow2d_oracle.so")
_lib = None

GREY, GRADIENT, GRADIENT_UNTILED, LOG_DERIVATIVES = 0, 1, 2, 3  # 2: true neighbours instead of the 16x8 tile rule (not a reference mode)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "flow2d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libflow2d_oracle.so"])
    return _LIB_PATH


class FlowParams(C.Structure):
    _fields_ = [
        ("warp_levels_count", C.c_size_t),
        ("warp_scale_factor", C.c_float),
        ("outer_iterations_count", C.c_size_t),
        ("inner_iterations_count", C.c_size_t),
        ("equation_alpha", C.c_float),
        ("equation_smoothness", C.c_float),
        ("equation_data", C.c_float),
        ("median_radius", C.c_size_t),
        ("gaussian_sigma", C.c_float),
        ("data_constancy", C.c_int),
        ("sor_omega", C.c_float),
    ]


DUMP_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p)


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota.  OpenMP's own
    default (all hardware threads) oversubscribes a quota-limited container badly."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        build()
        # idle OpenMP workers must sleep, not spin: the checker shares the box with the code under test
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        os.environ.setdefault("GOMP_SPINCOUNT", "0")
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        sz = C.c_size_t
        f = C.c_float
        L.oracle_max_threads.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_max_warp_level.restype = sz
        L.oracle_max_warp_level.argtypes = [sz, sz, f]
        L.oracle_level_geometry.argtypes = [sz, sz, f, C.c_int, C.POINTER(sz), C.POINTER(sz), fp, fp]
        L.oracle_gaussian_taps.restype = C.c_int
        L.oracle_gaussian_taps.argtypes = [f, fp, C.POINTER(C.c_int)]
        L.oracle_convolution_rows.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int]
        L.oracle_convolution_cols.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int]
        L.oracle_resample_x.argtypes = [fp, fp, sz, sz, sz, sz]
        L.oracle_resample_y.argtypes = [fp, fp, sz, sz, sz, sz]
        L.oracle_resample.argtypes = [fp, fp, fp, sz, sz, sz, sz, sz]
        L.oracle_registration_2d.argtypes = [fp, fp, fp, fp, sz, sz, sz, f, f, fp]
        L.oracle_compute_phi_ksi.argtypes = [fp] * 6 + [sz, sz, sz, f, f, f, f, fp, fp]
        L.oracle_solve_2d.argtypes = [fp] * 8 + [sz, sz, sz, f, f, f, fp, fp]
        L.oracle_solve_2d_grad.argtypes = [fp] * 8 + [sz, sz, sz, f, f, f, fp, fp]
        L.oracle_solve_2d_grad_untiled.argtypes = [fp] * 8 + [sz, sz, sz, f, f, f, fp, fp]
        L.oracle_solve_2d_log.argtypes = [fp] * 8 + [sz, sz, sz, f, f, f, fp, fp]
        L.oracle_solve_2d_log_planes.argtypes = [fp] * 8 + [sz, sz, sz, f, f, f, fp, fp]
        L.oracle_solve_2d_sor.argtypes = [fp] * 8 + [sz, sz, sz, f, f, f, f, C.c_int]
        L.oracle_solve_level_sor.argtypes = [fp] * 8 + [sz, sz, sz, sz, f, f, f, f, f, sz, sz, C.c_int, f]
        L.oracle_add_2d.argtypes = [fp, fp, sz, sz, sz]
        L.oracle_median_2d.restype = C.c_int
        L.oracle_median_2d.argtypes = [fp, sz, sz, sz, sz, fp]
        L.oracle_median_op.restype = C.c_int
        L.oracle_median_op.argtypes = [fp, sz, sz, sz, sz, sz, fp]
        L.oracle_solve_level.argtypes = [fp, fp, fp, fp, C.POINTER(fp), C.POINTER(fp), fp, fp, C.POINTER(fp),
                                         C.POINTER(fp), sz, sz, sz, sz, f, f, f, f, f, sz, sz, C.c_int]
        L.oracle_compute_flow.restype = C.c_int
        L.oracle_compute_flow.argtypes = [fp, fp, fp, fp, sz, sz, C.POINTER(FlowParams), DUMP_FN, C.c_void_p,
                                          C.POINTER(C.c_double)]
        _lib = L
        L.oracle_set_threads(min(usable_cpus(), 16))
    return _lib


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "float32 C-contiguous plane expected"
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pitch(a):
    return a.shape[1]


def max_threads():
    return lib().oracle_max_threads()


def set_threads(n):
    lib().oracle_set_threads(int(n))


def max_warp_level(width, height, scale):
    return int(lib().oracle_max_warp_level(width, height, scale))


def level_geometry(width, height, scale, level):
    lw, lh = C.c_size_t(), C.c_size_t()
    hx, hy = C.c_float(), C.c_float()
    lib().oracle_level_geometry(width, height, scale, level, C.byref(lw), C.byref(lh), C.byref(hx), C.byref(hy))
    return lw.value, lh.value, np.float32(hx.value), np.float32(hy.value)


def gaussian_taps(sigma):
    taps = np.zeros(51, np.float32)
    r = C.c_int()
    rc = lib().oracle_gaussian_taps(sigma, _p(taps), C.byref(r))
    if rc:
        raise ValueError("gaussian kernel longer than 51 taps")
    return taps[: 2 * r.value + 1].copy(), r.value


def convolution(src, w, h, sigma):
    """Rows then columns through a temp plane (cuda_operation_convolution_2d.cpp:169-175)."""
    taps, r = gaussian_taps(sigma)
    tmp = np.zeros_like(src)
    dst = np.zeros_like(src)
    lib().oracle_convolution_rows(_p(tmp), _p(src), w, h, _pitch(src), _p(taps), r)
    lib().oracle_convolution_cols(_p(dst), _p(tmp), w, h, _pitch(src), _p(taps), r)
    return dst


def convolution_rows(src, w, h, taps, r):
    dst = np.zeros_like(src)
    lib().oracle_convolution_rows(_p(dst), _p(src), w, h, _pitch(src), _p(np.ascontiguousarray(taps, np.float32)), r)
    return dst


def convolution_cols(src, w, h, taps, r):
    dst = np.zeros_like(src)
    lib().oracle_convolution_cols(_p(dst), _p(src), w, h, _pitch(src), _p(np.ascontiguousarray(taps, np.float32)), r)
    return dst


def resample_x(src, out_w, out_h, in_w):
    dst = np.zeros_like(src)
    lib().oracle_resample_x(_p(src), _p(dst), out_w, out_h, in_w, _pitch(src))
    return dst


def resample_y(src, out_w, out_h, in_h):
    dst = np.zeros_like(src)
    lib().oracle_resample_y(_p(src), _p(dst), out_w, out_h, in_h, _pitch(src))
    return dst


def resample(src, in_w, in_h, out_w, out_h):
    dst = np.zeros_like(src)
    tmp = np.zeros_like(src)
    lib().oracle_resample(_p(src), _p(dst), _p(tmp), in_w, in_h, out_w, out_h, _pitch(src))
    return dst


def registration(f0, f1, u, v, w, h, hx, hy):
    out = np.zeros_like(f0)
    lib().oracle_registration_2d(_p(f0), _p(f1), _p(u), _p(v), w, h, _pitch(f0), hx, hy, _p(out))
    return out


def compute_phi_ksi(f0, f1, u, v, du, dv, w, h, hx, hy, e_smooth, e_data):
    phi = np.zeros_like(f0)
    ksi = np.zeros_like(f0)
    lib().oracle_compute_phi_ksi(_p(f0), _p(f1), _p(u), _p(v), _p(du), _p(dv), w, h, _pitch(f0), hx, hy, e_smooth,
                                 e_data, _p(phi), _p(ksi))
    return phi, ksi


def solve_sweep(f0, f1, u, v, du, dv, phi, ksi, w, h, hx, hy, alpha, constancy=GREY):
    tdu = np.zeros_like(f0)
    tdv = np.zeros_like(f0)
    fn = {GREY: lib().oracle_solve_2d, GRADIENT: lib().oracle_solve_2d_grad,
          GRADIENT_UNTILED: lib().oracle_solve_2d_grad_untiled, LOG_DERIVATIVES: lib().oracle_solve_2d_log}[constancy]
    fn(_p(f0), _p(f1), _p(u), _p(v), _p(du), _p(dv), _p(phi), _p(ksi), w, h, _pitch(f0), hx, hy, alpha, _p(tdu),
       _p(tdv))
    return tdu, tdv


def solve_sweep_log_planes(lg0, lg1, u, v, du, dv, phi, ksi, w, h, hx, hy, alpha):
    """solve_2d_log with log(frame + 1) supplied by the caller (takes the libm out of a comparison)."""
    tdu = np.zeros_like(lg0)
    tdv = np.zeros_like(lg0)
    lib().oracle_solve_2d_log_planes(_p(lg0), _p(lg1), _p(u), _p(v), _p(du), _p(dv), _p(phi), _p(ksi), w, h,
                                     _pitch(lg0), hx, hy, alpha, _p(tdu), _p(tdv))
    return tdu, tdv


def sor_iteration(f0, f1, u, v, du, dv, phi, ksi, w, h, hx, hy, alpha, omega, constancy=GREY):
    """One opt-in red-black SOR iteration (not a reference kernel); returns the relaxed (du, dv)."""
    du, dv = du.copy(), dv.copy()
    lib().oracle_solve_2d_sor(_p(f0), _p(f1), _p(u), _p(v), _p(du), _p(dv), _p(phi), _p(ksi), w, h, _pitch(f0), hx, hy,
                              alpha, omega, constancy)
    return du, dv


def solve_level_sor(f0, f1, u, v, w, h, hx, hy, alpha, e_smooth, e_data, outer, inner, omega, constancy=GREY):
    du, dv, phi, ksi = (np.zeros_like(f0) for _ in range(4))
    lib().oracle_solve_level_sor(_p(f0), _p(f1), _p(u), _p(v), _p(du), _p(dv), _p(phi), _p(ksi), w, h, _pitch(f0),
                                 f0.shape[0], hx, hy, alpha, e_smooth, e_data, outer, inner, constancy, omega)
    return du, dv


def add(op0, op1, w, h):
    out = op0.copy()
    lib().oracle_add_2d(_p(out), _p(op1), w, h, _pitch(op0))
    return out


def median(src, w, h, radius):
    out = np.zeros_like(src)
    rc = lib().oracle_median_2d(_p(src), w, h, _pitch(src), radius, _p(out))
    if rc:
        raise ValueError("unsupported median width %d" % radius)
    return out


def solve_level(f0, f1, u, v, w, h, hx, hy, alpha, e_smooth, e_data, outer, inner, constancy=GREY):
    """Returns (du, dv, phi, ksi) after outer x inner sweeps (cuda_operation_solve_2d.cpp:229-300)."""
    bufs = [np.zeros_like(f0) for _ in range(6)]
    fp = C.POINTER(C.c_float)
    du, dv, tdu, tdv = (fp(), fp(), fp(), fp())
    du.contents, dv.contents = _p(bufs[0]).contents, _p(bufs[1]).contents
    tdu.contents, tdv.contents = _p(bufs[4]).contents, _p(bufs[5]).contents
    du, dv, tdu, tdv = _p(bufs[0]), _p(bufs[1]), _p(bufs[4]), _p(bufs[5])
    lib().oracle_solve_level(_p(f0), _p(f1), _p(u), _p(v), C.byref(du), C.byref(dv), _p(bufs[2]), _p(bufs[3]),
                             C.byref(tdu), C.byref(tdv), w, h, _pitch(f0), f0.shape[0], hx, hy, alpha, e_smooth,
                             e_data, outer, inner, constancy)
    addr = {b.ctypes.data: b for b in bufs}
    return (addr[C.addressof(du.contents)], addr[C.addressof(dv.contents)], bufs[2], bufs[3])


def compute_flow(frame_0, frame_1, levels, scale, outer, inner, alpha, e_smooth, e_data, median_radius, sigma,
                 constancy=GREY, dump=None, sor_omega=0.0):
    """Whole coarse-to-fine loop (optical_flow_2d.cpp:142-569) on tight H x W images.

    dump: optional callable(tag:str, level:int, plane:np.ndarray[h,w]) called per stage.
    Returns (u, v, finest_solve_seconds)."""
    f0 = np.ascontiguousarray(frame_0, np.float32)
    f1 = np.ascontiguousarray(frame_1, np.float32)
    H, W = f0.shape
    u = np.zeros((H, W), np.float32)
    v = np.zeros((H, W), np.float32)
    p = FlowParams(levels, scale, outer, inner, alpha, e_smooth, e_data, median_radius, sigma, constancy, sor_omega)

    def _cb(tag, level, ptr, w, h, pitch, _user):
        arr = np.ctypeslib.as_array(ptr, shape=(h * pitch,))[: (h - 1) * pitch + w]
        plane = np.empty((h, w), np.float32)
        for y in range(h):
            plane[y] = arr[y * pitch: y * pitch + w]
        dump(tag.decode(), level, plane)

    cb = DUMP_FN(_cb) if dump is not None else C.cast(None, DUMP_FN)
    t = C.c_double(0.0)
    rc = lib().oracle_compute_flow(_p(f0), _p(f1), _p(u), _p(v), W, H, C.byref(p), cb, None, C.byref(t))
    if rc:
        raise ValueError("oracle_compute_flow failed with status %d" % rc)
    return u, v, t.value


def synthetic_pair(width, height, dx, dy, seed=0, noise=False):
    """Deterministic translating-sinusoid pair, SURVEY.md section 8(d).

    I0(x,y) = 128 + 60 sin(2 pi x/64) cos(2 pi y/48) + 30 sin(2 pi (x+2y)/23.7); I1(x,y) = I0(x-dx, y-dy),
    evaluated in double, stored float32.  Optional uniform noise +-1 from a 64-bit LCG."""
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)

    def img(xx, yy):
        return (128.0 + 60.0 * np.sin(2 * np.pi * xx / 64.0) * np.cos(2 * np.pi * yy / 48.0)
                + 30.0 * np.sin(2 * np.pi * (xx + 2 * yy) / 23.7))

    i0 = img(x, y)
    i1 = img(x - dx, y - dy)
    if noise:
        n = width * height * 2
        state = np.uint64(seed * 2654435761 + 1442695040888963407 & 0xFFFFFFFFFFFFFFFF)
        out = np.empty(n, np.float64)
        a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
        with np.errstate(over="ignore"):
            for i in range(n):
                state = state * a + c
                out[i] = (int(state >> np.uint64(11)) / float(1 << 53)) * 2.0 - 1.0
        i0 = i0 + out[: n // 2].reshape(height, width)
        i1 = i1 + out[n // 2:].reshape(height, width)
    return i0.astype(np.float32), i1.astype(np.float32)
