"""ctypes front end of oracle/libref_driver.so: THE REFERENCE'S OWN KERNELS on the MI355X.

TEST INFRASTRUCTURE ONLY (tests/, tools/ and bench.py's optional reference leg) -- never imported by
the product package.  oracle/_ref/*.co are the reference's src/kernels/*_2d.cu compiled by hipcc for
gfx950 from the sources where they lie (oracle/Makefile, target ref-kernels; built in the container
that has /root/reference, prebuilt files travel to the GPU box).  ref_driver.cpp binds to them by
symbol name with each operator's launch geometry (cuda_operation_*_2d.cpp).

`RefKernels(cw, ch)` is one "container size" (the reference's full-resolution pitched planes); every
method takes level-sized (h, w) float32 arrays, places them in the top-left corner of containers whose
remaining area holds a sentinel, runs the reference kernel and returns the level-sized result.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libref_driver.so")
REF_DIR = os.path.join(_HERE, "_ref")
OPS = ("add", "convolution", "median", "registration", "resample", "solve")
GREY, GRADIENT, LOG_DERIVATIVES = 0, 1, 2  # data_structs.h:27

_lib = None


def available(fma=False):
    """True when the reference's kernels were built (oracle/_ref is git-ignored build output)."""
    d = os.path.join(REF_DIR, "fma") if fma else REF_DIR
    return all(os.path.exists(os.path.join(d, "%s_2d.co" % op)) for op in OPS)


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
    ]


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "ref_driver.cpp")
        if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libref_driver.so"])
        L = C.CDLL(LIB_PATH)
        fp, sz, f, i = C.POINTER(C.c_float), C.c_size_t, C.c_float, C.c_int
        pfp = C.POINTER(fp)
        L.refk_last_error.restype = C.c_char_p
        L.refk_open.argtypes = [C.c_char_p, sz, sz, sz]
        L.refk_plane_alloc.restype = fp
        L.refk_upload.argtypes = [fp, fp]
        L.refk_download.argtypes = [fp, fp]
        L.refk_add.argtypes = [fp, fp, sz, sz]
        L.refk_convolution_taps.argtypes = [fp, fp, fp, sz, sz, fp, i]
        L.refk_gaussian_taps.argtypes = [f, fp, C.POINTER(i)]
        L.refk_convolution.argtypes = [fp, fp, fp, sz, sz, f]
        L.refk_median.argtypes = [fp, fp, sz, sz, sz]
        L.refk_registration.argtypes = [fp] * 5 + [sz, sz, f, f]
        L.refk_resample.argtypes = [fp, fp, fp, sz, sz, sz, sz]
        L.refk_phi_ksi.argtypes = [fp] * 6 + [sz, sz, f, f, f, f, fp, fp]
        L.refk_sweep.argtypes = [i] + [fp] * 8 + [sz, sz, f, f, f, fp, fp]
        L.refk_solve.argtypes = [fp, fp, fp, fp, pfp, pfp, fp, fp, pfp, pfp, sz, sz, f, f, i, sz, sz, f, f, f, fp]
        L.refk_compute_flow.argtypes = [fp, fp, fp, fp, C.POINTER(FlowParams), fp, fp]
        _lib = L
    return _lib


def _hp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


class RefKernels:
    SENTINEL = 777.0

    def __init__(self, cw, ch, fma=False, pitch_bytes=None):
        if not available(fma):
            raise RuntimeError("oracle/_ref has no reference kernels (run `make -C oracle ref` where "
                               "/root/reference exists)")
        self.L = lib()
        self.cw, self.ch = int(cw), int(ch)
        self.pitch_bytes = int(pitch_bytes) if pitch_bytes else -(-self.cw * 4 // 512) * 512  # cuMemAllocPitch-like
        self.pitch = self.pitch_bytes // 4
        d = os.path.join(REF_DIR, "fma") if fma else REF_DIR
        self._check(self.L.refk_open(d.encode(), self.cw, self.ch, self.pitch_bytes))
        self._free = []

    def _check(self, rc):
        if rc:
            raise RuntimeError("reference driver: status %d: %s" % (rc, self.L.refk_last_error().decode()))

    def close(self):
        self.L.refk_close()
        self._free = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- planes --------------------------------------------------------------------------------
    def _plane(self, level=None, fill=None):
        p = self._free.pop() if self._free else self.L.refk_plane_alloc()
        if not p:
            raise MemoryError("refk_plane_alloc")
        host = np.full((self.ch, self.pitch), self.SENTINEL if fill is None else fill, np.float32)
        if level is not None:
            host[: level.shape[0], : level.shape[1]] = level
        self._check(self.L.refk_upload(p, _hp(host)))
        return p

    def _get(self, p, w=None, h=None):
        host = np.empty((self.ch, self.pitch), np.float32)
        self._check(self.L.refk_download(p, _hp(host)))
        return host if w is None else host[:h, :w].copy()

    def _release(self, *planes):
        self._free.extend(planes)

    # -- operators -----------------------------------------------------------------------------
    def add(self, a, b):
        h, w = a.shape
        pa, pb = self._plane(a), self._plane(b)
        self._check(self.L.refk_add(pa, pb, w, h))
        full = self._get(pa)
        self._release(pa, pb)
        return full[:h, :w].copy(), full

    def gaussian_taps(self, sigma):
        taps = np.zeros(51, np.float32)
        r = C.c_int()
        if self.L.refk_gaussian_taps(sigma, _hp(taps), C.byref(r)):
            raise ValueError("gaussian kernel longer than 51 taps")
        return taps[: 2 * r.value + 1].copy(), r.value

    def convolution(self, src, sigma, taps=None, radius=None):
        h, w = src.shape
        pi, po, pt = self._plane(src, 0.0), self._plane(), self._plane()
        if taps is None:
            rc = self.L.refk_convolution(pi, po, pt, w, h, sigma)
        else:
            rc = self.L.refk_convolution_taps(pi, po, pt, w, h, _hp(np.ascontiguousarray(taps, np.float32)), radius)
        self._check(rc)
        out = self._get(po, w, h)
        self._release(pi, po, pt)
        return out

    def median(self, src, radius):
        """Returns (status, output level, output container); status 4 = refused, nothing written."""
        h, w = src.shape
        pi, po = self._plane(src), self._plane()
        rc = self.L.refk_median(pi, po, w, h, radius)
        if rc == 1:
            self._check(rc)
        full = self._get(po)
        self._release(pi, po)
        return rc, full[:h, :w].copy(), full

    def registration(self, f0, f1, u, v, hx, hy):
        h, w = f0.shape
        p = [self._plane(a) for a in (f0, f1, u, v)]
        po = self._plane()
        self._check(self.L.refk_registration(*p, po, w, h, hx, hy))
        out = self._get(po, w, h)
        self._release(po, *p)
        return out

    def resample(self, src, rw, rh):
        h, w = src.shape
        pi, po, pt = self._plane(src), self._plane(), self._plane()
        self._check(self.L.refk_resample(pi, po, pt, w, h, rw, rh))
        out = self._get(po, rw, rh)
        self._release(pi, po, pt)
        return out

    def phi_ksi(self, f0, f1, u, v, du, dv, hx, hy, e_smooth, e_data):
        h, w = f0.shape
        p = [self._plane(a) for a in (f0, f1, u, v, du, dv)]
        phi, ksi = self._plane(), self._plane()
        self._check(self.L.refk_phi_ksi(*p, w, h, hx, hy, e_smooth, e_data, phi, ksi))
        out = self._get(phi, w, h), self._get(ksi, w, h)
        self._release(phi, ksi, *p)
        return out

    def sweep(self, constancy, f0, f1, u, v, du, dv, phi, ksi, hx, hy, alpha):
        h, w = f0.shape
        p = [self._plane(a) for a in (f0, f1, u, v, du, dv, phi, ksi)]
        tdu, tdv = self._plane(), self._plane()
        self._check(self.L.refk_sweep(constancy, *p, w, h, hx, hy, alpha, tdu, tdv))
        out = self._get(tdu, w, h), self._get(tdv, w, h)
        self._release(tdu, tdv, *p)
        return out

    def solve(self, f0, f1, u, v, hx, hy, constancy, outer, inner, alpha, e_smooth, e_data):
        """CudaOperationSolve2D::Execute; returns (du, dv, phi, ksi, milliseconds)."""
        h, w = f0.shape
        p = [self._plane(a) for a in (f0, f1, u, v)]
        fp = C.POINTER(C.c_float)
        du, dv, tdu, tdv = (self._plane() for _ in range(4))
        phi, ksi = self._plane(), self._plane()
        cdu, cdv, ctdu, ctdv = (C.cast(x, fp) for x in (du, dv, tdu, tdv))
        ms = C.c_float()
        self._check(self.L.refk_solve(*p, C.byref(cdu), C.byref(cdv), phi, ksi, C.byref(ctdu), C.byref(ctdv), w, h, hx,
                                      hy, constancy, outer, inner, alpha, e_smooth, e_data, C.byref(ms)))
        out = (self._get(cdu, w, h), self._get(cdv, w, h), self._get(phi, w, h), self._get(ksi, w, h), ms.value)
        self._release(du, dv, tdu, tdv, phi, ksi, *p)
        return out

    def compute_flow(self, frame_0, frame_1, levels, scale, outer, inner, alpha, e_smooth, e_data, median_radius,
                     sigma, constancy=GREY):
        """ComputeFlow over the reference's kernels on tight (ch, cw) images -> (u, v, total ms, finest solve ms)."""
        f0 = np.ascontiguousarray(frame_0, np.float32)
        f1 = np.ascontiguousarray(frame_1, np.float32)
        assert f0.shape == (self.ch, self.cw) == f1.shape
        u = np.zeros_like(f0)
        v = np.zeros_like(f0)
        p = FlowParams(levels, scale, outer, inner, alpha, e_smooth, e_data, median_radius, sigma, constancy)
        total, finest = C.c_float(), C.c_float()
        self._check(self.L.refk_compute_flow(_hp(f0), _hp(f1), _hp(u), _hp(v), C.byref(p), C.byref(total),
                                             C.byref(finest)))
        return u, v, total.value, finest.value
