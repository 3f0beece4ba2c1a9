"""Python plumbing for the MI355X-native flow2d hot path (ctypes over the C-ABI).

The product is the HIP library (csrc/ -> libflow2d_hip.so, include/flow2d_c_abi.h) and the C++ host
layer (host/ -> libflow2d_host.so, `flow2d` CLI) that mirrors the reference's OpticalFlow2D /
CudaOperation* interface.  This module only loads those libraries for tests/ and bench.py; it holds
no algorithm and has NO CPU fallback: a missing library or a failing call raises.

The directory name carries a hyphen, so import it with
    importlib.import_module("cuda-flow2d_amd")
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
# FLOW2D_HIP_LIB: an experimental build of the HIP library (ab/*.so, tools/ab_time.sh) instead of the in-tree one
HIP_LIB_PATH = os.environ.get("FLOW2D_HIP_LIB") or os.path.join(_HERE, "csrc", "libflow2d_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "host", "libflow2d_host.so")
CLI_PATH = os.path.join(_HERE, "host", "flow2d")

# flow2d_constancy; 2 = true-neighbour gradient term (not in the reference), 3 = the reference's LogDerivatives
GREY, GRADIENT, GRADIENT_UNTILED, LOG_DERIVATIVES = 0, 1, 2, 3
_HOST_CONSTANCY = {GREY: 0, GRADIENT: 1, GRADIENT_UNTILED: 3, LOG_DERIVATIVES: 2}  # enum class DataConstancy
SOLVER_AUTO, SOLVER_PER_SWEEP, SOLVER_FUSED, SOLVER_SINGLE_WORKGROUP, SOLVER_TILED = 0, 1, 2, 3, 4

STATUS = {0: "ok", 1: "invalid argument", 2: "no usable HIP device", 3: "HIP runtime error",
          4: "out of device memory", 5: "unsupported parameter"}


class Flow2DError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s failed: status %d (%s)%s" % (where, status, STATUS.get(status, "?"),
                                                          (": " + detail) if detail else ""))


def build(jobs=8):
    """Compile every native piece in-tree: HIP kernels + C-ABI (hipcc, gfx950) and the C++ host layer."""
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", os.path.join(_HERE, "csrc")])
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", os.path.join(_HERE, "host")])


class SolveParams(C.Structure):
    _fields_ = [
        ("width", C.c_size_t), ("height", C.c_size_t), ("pitch_bytes", C.c_size_t),
        ("container_height", C.c_size_t), ("hx", C.c_float), ("hy", C.c_float),
        ("equation_alpha", C.c_float), ("equation_smoothness", C.c_float), ("equation_data", C.c_float),
        ("outer_iterations_count", C.c_size_t), ("inner_iterations_count", C.c_size_t),
        ("data_constancy", C.c_int), ("algorithm", C.c_int), ("sor_omega", C.c_float),
    ]


class TimingRecord(C.Structure):
    _fields_ = [
        ("width", C.c_size_t), ("height", C.c_size_t), ("outer", C.c_size_t), ("inner", C.c_size_t),
        ("data_constancy", C.c_int), ("algorithm", C.c_int), ("kernel_launches", C.c_int),
        ("elapsed_ms", C.c_float), ("kernel_ms", C.c_float), ("algorithmic_bytes_per_launch", C.c_double),
    ]


_hip = None


def hip_lib():
    """The C-ABI library.  Raises if it has not been built -- there is no fallback."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP extension is mandatory, there is no CPU path)" % HIP_LIB_PATH)
        L = C.CDLL(HIP_LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, sz, f, i = C.c_void_p, C.c_size_t, C.c_float, C.c_int
        L.flow2d_abi_version.restype = i
        L.flow2d_status_string.restype = C.c_char_p
        L.flow2d_status_string.argtypes = [i]
        L.flow2d_last_error.restype = C.c_char_p
        L.flow2d_device_count.argtypes = [C.POINTER(i)]
        L.flow2d_hw_queues.restype = i
        L.flow2d_request_hw_queues.argtypes = [i]
        # the lanes of the batched path want a hardware queue each; a refusal (HIP already running, e.g. under torch: bench.py
        # exports the variable itself) is reported by flow2d_last_error() and OpticalFlowBatch2D's warning, not here
        L.flow2d_request_hw_queues(8)
        L.flow2d_context_create.argtypes = [i, C.POINTER(vp)]
        L.flow2d_context_create_on_stream.argtypes = [i, vp, C.POINTER(vp)]
        L.flow2d_context_destroy.argtypes = [vp]
        L.flow2d_context_device.argtypes = [vp, C.POINTER(i)]
        L.flow2d_context_stream.argtypes = [vp, C.POINTER(vp)]
        L.flow2d_synchronize.argtypes = [vp]
        L.flow2d_mem_info.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
        L.flow2d_device_name.argtypes = [vp, C.c_char_p, sz]
        L.flow2d_plane_pitch_bytes.restype = sz
        L.flow2d_plane_pitch_bytes.argtypes = [sz]
        L.flow2d_plane_alloc.argtypes = [vp, sz, sz, C.POINTER(vp), C.POINTER(sz)]
        L.flow2d_plane_free.argtypes = [vp, vp]
        L.flow2d_memset_2d.argtypes = [vp, vp, sz, i, sz, sz]
        L.flow2d_copy_h2d_2d.argtypes = [vp, vp, sz, vp, sz, sz, sz]
        L.flow2d_copy_d2h_2d.argtypes = [vp, vp, sz, vp, sz, sz, sz]
        L.flow2d_copy_d2d.argtypes = [vp, vp, vp, sz]
        L.flow2d_copy_planes.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(vp), sz, sz, sz]
        L.flow2d_event_create.argtypes = [vp, C.POINTER(vp)]
        L.flow2d_event_record.argtypes = [vp, vp]
        L.flow2d_event_synchronize.argtypes = [vp, vp]
        L.flow2d_event_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(f)]
        L.flow2d_event_destroy.argtypes = [vp, vp]
        L.flow2d_stream_wait_event.argtypes = [vp, vp]
        L.flow2d_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
        L.flow2d_host_free.argtypes = [vp, vp]
        L.flow2d_add_2d.argtypes = [vp, vp, vp, sz, sz, sz]
        L.flow2d_gaussian_kernel.argtypes = [f, C.POINTER(f), C.POINTER(i)]
        L.flow2d_convolution_rows.argtypes = [vp, vp, vp, sz, sz, sz, C.POINTER(f), i]
        L.flow2d_convolution_columns.argtypes = [vp, vp, vp, sz, sz, sz, C.POINTER(f), i]
        L.flow2d_gaussian_blur.argtypes = [vp, vp, vp, sz, sz, sz, C.POINTER(f), i]
        L.flow2d_median_2d.argtypes = [vp, vp, sz, sz, sz, sz, vp]
        L.flow2d_registration_2d.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, f, f, vp]
        L.flow2d_upsample_registration_2d.argtypes = [vp, vp, vp, sz, sz, vp, vp, vp, vp, sz, sz, sz, f, f, vp]
        L.flow2d_resample_x.argtypes = [vp, vp, vp, sz, sz, sz, sz]
        L.flow2d_resample_y.argtypes = [vp, vp, vp, sz, sz, sz, sz]
        L.flow2d_add_2d_pair.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz]
        L.flow2d_median_2d_pair.argtypes = [vp, vp, vp, sz, sz, sz, sz, vp, vp]
        L.flow2d_add_median_2d_pair.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz, vp, vp]
        L.flow2d_resample_x_pair.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz]
        L.flow2d_resample_y_pair.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz]
        L.flow2d_resample_xy_pair.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz, sz]
        L.flow2d_resample_x_levels.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, sz, C.POINTER(sz), C.POINTER(sz)]
        L.flow2d_compute_phi_ksi.argtypes = [vp] * 7 + [sz, sz, sz, f, f, f, f, vp, vp]
        L.flow2d_solve_2d.argtypes = [vp] * 9 + [sz, sz, sz, f, f, f, vp, vp]
        L.flow2d_solve_2d_grad.argtypes = [vp] * 9 + [sz, sz, sz, f, f, f, vp, vp]
        L.flow2d_solve_2d_grad_untiled.argtypes = [vp] * 9 + [sz, sz, sz, f, f, f, vp, vp]
        L.flow2d_solve_2d_log.argtypes = [vp] * 9 + [sz, sz, sz, f, f, f, vp, vp]
        L.flow2d_solver_algorithm_for.argtypes = [i, sz, sz, sz, sz, sz, i]
        L.flow2d_solve_2d_sor.argtypes = [vp] * 9 + [sz, sz, sz, f, f, f, f, i]
        L.flow2d_solve_level.argtypes = [vp] * 11 + [C.POINTER(SolveParams), C.POINTER(i)]
        L.flow2d_timing_enable.argtypes = [vp, i]
        L.flow2d_fused_fallbacks.argtypes = [vp, C.POINTER(C.c_ulonglong)]
        L.flow2d_context_set_lone.argtypes = [vp, i]
        L.flow2d_resample_y_levels.argtypes = [vp, vp, vp, vp, vp, sz, sz, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
        L.flow2d_clock_probe_start.argtypes = [vp, C.c_double]
        L.flow2d_clock_probe_read.argtypes = [vp, C.POINTER(C.c_double)]
        L.flow2d_fused_block_order.argtypes = [vp, sz, sz, sz, sz, C.POINTER(i), sz, C.POINTER(sz)]
        if hasattr(L, "flow2d_fused_plain_waves"):  # (absent from libraries of earlier rounds loaded for an A/B)
            L.flow2d_fused_plain_waves.argtypes = [vp, C.POINTER(C.c_ulonglong)]
        L.flow2d_timing_launch_filter.argtypes = [vp, sz, sz]
        L.flow2d_timing_count.argtypes = [vp, C.POINTER(sz)]
        L.flow2d_timing_get.argtypes = [vp, sz, C.POINTER(TimingRecord)]
        L.flow2d_timing_reset.argtypes = [vp]
        _hip = L
    return _hip


def _check(status, where):
    if status != 0:
        raise Flow2DError(status, where, hip_lib().flow2d_last_error().decode(errors="replace"))


def device_count():
    n = C.c_int(0)
    st = hip_lib().flow2d_device_count(C.byref(n))
    return n.value if st == 0 else 0


def gaussian_kernel(sigma):
    taps = (C.c_float * 51)()
    r = C.c_int(0)
    _check(hip_lib().flow2d_gaussian_kernel(sigma, taps, C.byref(r)), "flow2d_gaussian_kernel")
    return np.array(taps[: 2 * r.value + 1], np.float32), r.value


class Plane:
    """A pitched fp32 container in HBM (reference: one of the 12 cuMemAllocPitch planes)."""

    def __init__(self, ctx, width, height):
        self.ctx = ctx
        self.width, self.height = width, height
        ptr, pitch = C.c_void_p(), C.c_size_t()
        _check(hip_lib().flow2d_plane_alloc(ctx.handle, width, height, C.byref(ptr), C.byref(pitch)),
               "flow2d_plane_alloc")
        self.ptr, self.pitch = ptr.value, pitch.value

    def upload(self, array):
        a = np.ascontiguousarray(array, np.float32)
        h, w = a.shape
        assert w <= self.width and h <= self.height
        _check(hip_lib().flow2d_copy_h2d_2d(self.ctx.handle, self.ptr, self.pitch, a.ctypes.data, w * 4, w * 4, h),
               "flow2d_copy_h2d_2d")
        self.ctx.synchronize()
        return self

    def download(self, width=None, height=None):
        w = self.width if width is None else width
        h = self.height if height is None else height
        out = np.empty((h, w), np.float32)
        _check(hip_lib().flow2d_copy_d2h_2d(self.ctx.handle, out.ctypes.data, w * 4, self.ptr, self.pitch, w * 4, h),
               "flow2d_copy_d2h_2d")
        self.ctx.synchronize()
        return out

    def fill_bytes(self, value=0):
        _check(hip_lib().flow2d_memset_2d(self.ctx.handle, self.ptr, self.pitch, value, self.width * 4, self.height),
               "flow2d_memset_2d")
        return self

    def free(self):
        if self.ptr:
            hip_lib().flow2d_plane_free(self.ctx.handle, self.ptr)
            self.ptr = None


class Context:
    """One device + stream (reference: the CUcontext of main.cpp:51 plus the NULL stream)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        if stream is None:
            _check(hip_lib().flow2d_context_create(device, C.byref(h)), "flow2d_context_create")
        else:
            _check(hip_lib().flow2d_context_create_on_stream(device, stream, C.byref(h)),
                   "flow2d_context_create_on_stream")
        self.handle = h
        self._planes = []

    # -- memory ---------------------------------------------------------------------------------
    def plane(self, width, height, data=None):
        p = Plane(self, width, height)
        self._planes.append(p)
        if data is not None:
            p.fill_bytes(0)
            p.upload(data)
        return p

    def copy_planes(self, srcs, dsts, width, height):
        """flow2d_copy_planes: the planes srcs[i] -> dsts[i] (same pitch) with one launch."""
        n = len(srcs)
        a, b = (C.c_void_p * n)(*[p.ptr for p in srcs]), (C.c_void_p * n)(*[p.ptr for p in dsts])
        _check(hip_lib().flow2d_copy_planes(self.handle, n, a, b, srcs[0].pitch, width, height), "flow2d_copy_planes")

    def synchronize(self):
        _check(hip_lib().flow2d_synchronize(self.handle), "flow2d_synchronize")

    def mem_info(self):
        a, b = C.c_size_t(), C.c_size_t()
        _check(hip_lib().flow2d_mem_info(self.handle, C.byref(a), C.byref(b)), "flow2d_mem_info")
        return a.value, b.value

    def fused_fallbacks(self):
        """Waves of the fused kernel that repeated their strip with the plain division (synchronises)."""
        n = C.c_ulonglong()
        _check(hip_lib().flow2d_fused_fallbacks(self.handle, C.byref(n)), "flow2d_fused_fallbacks")
        return n.value

    def set_lone(self, lone):
        """flow2d_context_set_lone: this context's launches run alone on the device (packed strip build for under-filled launches)"""
        _check(hip_lib().flow2d_context_set_lone(self.handle, int(bool(lone))), "flow2d_context_set_lone")

    def resample_y_levels(self, packed_a, out_a, in_height, widths, heights, columns, rows, packed_b=None, out_b=None):
        """the y passes of several pyramid levels in one launch (flow2d_resample_y_levels)"""
        n = len(widths)
        arr = lambda v: (C.c_size_t * n)(*v)
        _check(hip_lib().flow2d_resample_y_levels(self.handle, packed_a.ptr, out_a.ptr, packed_b.ptr if packed_b else None,
                                                  out_b.ptr if out_b else None, in_height, packed_a.pitch, n, arr(widths), arr(heights),
                                                  arr(columns), arr(rows)), "flow2d_resample_y_levels")

    def clock_probe_start(self, duration_us):
        """queues one sleeping wave per XCD on this context's stream that brackets duration_us with the 100 MHz and the shader clock"""
        _check(hip_lib().flow2d_clock_probe_start(self.handle, float(duration_us)), "flow2d_clock_probe_start")

    def clock_probe_read(self):
        """waits for the probe; the shader clock held per XCD in GHz (0: no wave landed there)"""
        ghz = (C.c_double * 8)()
        _check(hip_lib().flow2d_clock_probe_read(self.handle, ghz), "flow2d_clock_probe_read")
        return list(ghz)

    def fused_block_order(self, width, height, inner, instances=1):
        """(grid, 4) int array: block column, strip, first row, end row of every launch block of a strip launch (-1: empty id)."""
        grid = C.c_size_t(0)
        cap = 1 << 16
        buf = (C.c_int * (4 * cap))()
        _check(hip_lib().flow2d_fused_block_order(self.handle, width, height, inner, instances, buf, cap, C.byref(grid)),
               "flow2d_fused_block_order")
        return np.frombuffer(buf, np.int32, 4 * grid.value).reshape(-1, 4).copy()

    def fused_plain_waves(self):
        """Waves of fused launches that ran the plain expressions throughout (grid spacing outside the proven range)."""
        n = C.c_ulonglong()
        _check(hip_lib().flow2d_fused_plain_waves(self.handle, C.byref(n)), "flow2d_fused_plain_waves")
        return n.value

    def device_name(self):
        buf = C.create_string_buffer(256)
        _check(hip_lib().flow2d_device_name(self.handle, buf, 256), "flow2d_device_name")
        return buf.value.decode()

    def close(self):
        if self.handle:
            for p in self._planes:
                p.free()
            self._planes = []
            hip_lib().flow2d_context_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- events ---------------------------------------------------------------------------------
    def event(self):
        e = C.c_void_p()
        _check(hip_lib().flow2d_event_create(self.handle, C.byref(e)), "flow2d_event_create")
        return e

    def record(self, ev):
        _check(hip_lib().flow2d_event_record(self.handle, ev), "flow2d_event_record")

    def wait_event(self, ev):
        """Host wait until everything recorded before `ev` on this context's stream has finished."""
        _check(hip_lib().flow2d_event_synchronize(self.handle, ev), "flow2d_event_synchronize")

    def elapsed_ms(self, start, stop):
        _check(hip_lib().flow2d_event_synchronize(self.handle, stop), "flow2d_event_synchronize")
        ms = C.c_float()
        _check(hip_lib().flow2d_event_elapsed_ms(self.handle, start, stop, C.byref(ms)), "flow2d_event_elapsed_ms")
        return ms.value

    # -- kernels (thin 1:1 wrappers of the C-ABI launchers) --------------------------------------
    def add(self, op0, op1, w, h):
        _check(hip_lib().flow2d_add_2d(self.handle, op0.ptr, op1.ptr, w, h, op0.pitch), "flow2d_add_2d")

    def convolution_rows(self, dst, src, w, h, taps, radius):
        t = np.ascontiguousarray(taps, np.float32)
        _check(hip_lib().flow2d_convolution_rows(self.handle, dst.ptr, src.ptr, w, h, src.pitch,
                                                 t.ctypes.data_as(C.POINTER(C.c_float)), radius),
               "flow2d_convolution_rows")

    def convolution_columns(self, dst, src, w, h, taps, radius):
        t = np.ascontiguousarray(taps, np.float32)
        _check(hip_lib().flow2d_convolution_columns(self.handle, dst.ptr, src.ptr, w, h, src.pitch,
                                                    t.ctypes.data_as(C.POINTER(C.c_float)), radius),
               "flow2d_convolution_columns")

    def gaussian_blur(self, dst, src, w, h, taps, radius):
        t = np.ascontiguousarray(taps, np.float32)
        _check(hip_lib().flow2d_gaussian_blur(self.handle, dst.ptr, src.ptr, w, h, src.pitch,
                                              t.ctypes.data_as(C.POINTER(C.c_float)), radius), "flow2d_gaussian_blur")

    def median(self, src, w, h, window, dst):
        _check(hip_lib().flow2d_median_2d(self.handle, src.ptr, w, h, src.pitch, window, dst.ptr), "flow2d_median_2d")

    def registration(self, f0, f1, u, v, w, h, hx, hy, out):
        _check(hip_lib().flow2d_registration_2d(self.handle, f0.ptr, f1.ptr, u.ptr, v.ptr, w, h, f0.pitch, hx, hy,
                                                out.ptr), "flow2d_registration_2d")

    def resample_x(self, src, dst, out_w, out_h, in_w):
        _check(hip_lib().flow2d_resample_x(self.handle, src.ptr, dst.ptr, out_w, out_h, in_w, src.pitch),
               "flow2d_resample_x")

    def resample_y(self, src, dst, out_w, out_h, in_h):
        _check(hip_lib().flow2d_resample_y(self.handle, src.ptr, dst.ptr, out_w, out_h, in_h, src.pitch),
               "flow2d_resample_y")

    def resample_xy(self, src, dst, in_w, in_h, out_w, out_h, src_b=None, dst_b=None):
        """Both resample passes in one launch (no temp plane); optional second plane."""
        _check(hip_lib().flow2d_resample_xy_pair(self.handle, src.ptr, dst.ptr, src_b.ptr if src_b else None,
                                                 dst_b.ptr if dst_b else None, in_w, in_h, out_w, out_h, src.pitch),
               "flow2d_resample_xy_pair")

    def upsample_registration(self, u, v, in_w, in_h, out_u, out_v, f0, f1, w, h, hx, hy, out):
        """(u, v) of the previous level resampled to w x h into out_u / out_v and f1 warped by them into `out`: one launch.
        u = v = None with in_w = in_h = 0 (the coarsest level): out_u = out_v = 0 and the warp by that."""
        _check(hip_lib().flow2d_upsample_registration_2d(self.handle, u.ptr if u else None, v.ptr if v else None, in_w, in_h, out_u.ptr,
                                                         out_v.ptr, f0.ptr, f1.ptr,
                                                         w, h, f0.pitch, hx, hy, out.ptr), "flow2d_upsample_registration_2d")

    def resample_x_levels(self, src, packed, in_w, h, widths, columns, src_b=None, packed_b=None):
        """x pass for several output widths in one trip over `src`; level l lands in columns[l] .. of `packed`."""
        n = len(widths)
        ws = (C.c_size_t * n)(*widths)
        cs = (C.c_size_t * n)(*columns)
        _check(hip_lib().flow2d_resample_x_levels(self.handle, src.ptr, packed.ptr, src_b.ptr if src_b else None,
                                                  packed_b.ptr if packed_b else None, in_w, h, src.pitch, n, ws, cs),
               "flow2d_resample_x_levels")

    # two planes of the same geometry per launch
    def add_pair(self, op0_a, op1_a, op0_b, op1_b, w, h):
        _check(hip_lib().flow2d_add_2d_pair(self.handle, op0_a.ptr, op1_a.ptr, op0_b.ptr, op1_b.ptr, w, h, op0_a.pitch),
               "flow2d_add_2d_pair")

    def median_pair(self, src_a, src_b, w, h, window, dst_a, dst_b):
        _check(hip_lib().flow2d_median_2d_pair(self.handle, src_a.ptr, src_b.ptr, w, h, src_a.pitch, window, dst_a.ptr,
                                               dst_b.ptr), "flow2d_median_2d_pair")

    def add_median(self, src_a, add_a, w, h, window, dst_a, src_b=None, add_b=None, dst_b=None):
        """Median of (src + add), the sum formed on the fly; optional second plane set."""
        _check(hip_lib().flow2d_add_median_2d_pair(self.handle, src_a.ptr, add_a.ptr, src_b.ptr if src_b else None,
                                                   add_b.ptr if add_b else None, w, h, src_a.pitch, window, dst_a.ptr,
                                                   dst_b.ptr if dst_b else None), "flow2d_add_median_2d_pair")

    def resample_x_pair(self, src_a, dst_a, src_b, dst_b, out_w, out_h, in_w):
        _check(hip_lib().flow2d_resample_x_pair(self.handle, src_a.ptr, dst_a.ptr, src_b.ptr, dst_b.ptr, out_w, out_h,
                                                in_w, src_a.pitch), "flow2d_resample_x_pair")

    def resample_y_pair(self, src_a, dst_a, src_b, dst_b, out_w, out_h, in_h):
        _check(hip_lib().flow2d_resample_y_pair(self.handle, src_a.ptr, dst_a.ptr, src_b.ptr, dst_b.ptr, out_w, out_h,
                                                in_h, src_a.pitch), "flow2d_resample_y_pair")

    def compute_phi_ksi(self, f0, f1, u, v, du, dv, w, h, hx, hy, e_smooth, e_data, phi, ksi):
        _check(hip_lib().flow2d_compute_phi_ksi(self.handle, f0.ptr, f1.ptr, u.ptr, v.ptr, du.ptr, dv.ptr, w, h,
                                                f0.pitch, hx, hy, e_smooth, e_data, phi.ptr, ksi.ptr),
               "flow2d_compute_phi_ksi")

    def solve_sweep(self, f0, f1, u, v, du, dv, phi, ksi, w, h, hx, hy, alpha, tdu, tdv, constancy=GREY):
        fn = {GREY: hip_lib().flow2d_solve_2d, GRADIENT: hip_lib().flow2d_solve_2d_grad,
              GRADIENT_UNTILED: hip_lib().flow2d_solve_2d_grad_untiled,
              LOG_DERIVATIVES: hip_lib().flow2d_solve_2d_log}[constancy]
        _check(fn(self.handle, f0.ptr, f1.ptr, u.ptr, v.ptr, du.ptr, dv.ptr, phi.ptr, ksi.ptr, w, h, f0.pitch, hx, hy,
                  alpha, tdu.ptr, tdv.ptr), "flow2d_solve_2d*")

    def sor_iteration(self, f0, f1, u, v, du, dv, phi, ksi, w, h, hx, hy, alpha, omega, constancy=GREY):
        _check(hip_lib().flow2d_solve_2d_sor(self.handle, f0.ptr, f1.ptr, u.ptr, v.ptr, du.ptr, dv.ptr, phi.ptr, ksi.ptr,
                                             w, h, f0.pitch, hx, hy, alpha, omega, constancy), "flow2d_solve_2d_sor")

    def solve_level(self, f0, f1, u, v, du, dv, phi, ksi, tdu, tdv, w, h, hx, hy, alpha, e_smooth, e_data, outer,
                    inner, constancy=GREY, algorithm=SOLVER_AUTO, container_height=None, sor_omega=0.0):
        """Returns (du_plane, dv_plane) holding the result (the library owns the ping-pong)."""
        p = SolveParams(w, h, f0.pitch, container_height or f0.height, hx, hy, alpha, e_smooth, e_data, outer, inner,
                        constancy, algorithm, sor_omega)
        flag = C.c_int(0)
        _check(hip_lib().flow2d_solve_level(self.handle, f0.ptr, f1.ptr, u.ptr, v.ptr, du.ptr, dv.ptr, phi.ptr,
                                            ksi.ptr, tdu.ptr, tdv.ptr, C.byref(p), C.byref(flag)),
               "flow2d_solve_level")
        return (tdu, tdv) if flag.value else (du, dv)

    # -- timing -----------------------------------------------------------------------------------
    def timing_enable(self, mode=1):
        _check(hip_lib().flow2d_timing_enable(self.handle, int(mode)), "flow2d_timing_enable")

    def timing_records(self):
        n = C.c_size_t()
        _check(hip_lib().flow2d_timing_count(self.handle, C.byref(n)), "flow2d_timing_count")
        out = []
        for k in range(n.value):
            r = TimingRecord()
            _check(hip_lib().flow2d_timing_get(self.handle, k, C.byref(r)), "flow2d_timing_get")
            out.append(r)
        return out

    def timing_reset(self):
        _check(hip_lib().flow2d_timing_reset(self.handle), "flow2d_timing_reset")


# ---- C++ host layer (OpticalFlow2D & friends) through its C facade ---------------------------------

class HostParams(C.Structure):
    _fields_ = [
        ("warp_levels_count", C.c_size_t), ("warp_scale_factor", C.c_float),
        ("outer_iterations_count", C.c_size_t), ("inner_iterations_count", C.c_size_t),
        ("equation_alpha", C.c_float), ("equation_smoothness", C.c_float), ("equation_data", C.c_float),
        ("median_radius", C.c_size_t), ("gaussian_sigma", C.c_float), ("solver_algorithm", C.c_int),
        ("sor_omega", C.c_float),
    ]


class HostSettings(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("medianRadius", C.c_int), ("iterInner", C.c_int),
        ("iterOuter", C.c_int), ("levels", C.c_int), ("press_key", C.c_int),
        ("sigma", C.c_float), ("alpha", C.c_float), ("e_smooth", C.c_float), ("e_data", C.c_float),
        ("warpScale", C.c_float),
        ("inputPath", C.c_char * 512), ("outputPath", C.c_char * 512), ("fileName1", C.c_char * 256),
        ("fileName2", C.c_char * 256), ("imageType", C.c_char * 32), ("dataConstancy", C.c_char * 32),
    ]


_host = None


def host_lib():
    """The C++ host layer.  Raises if it has not been built."""
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError("%s is missing: run __graft_entry__.build()" % HOST_LIB_PATH)
        L = C.CDLL(HOST_LIB_PATH)
        vp, sz, f, i = C.c_void_p, C.c_size_t, C.c_float, C.c_int
        fp = C.POINTER(C.c_float)
        L.flow2d_host_init_device.argtypes = [i]
        L.flow2d_host_adopt_context.argtypes = [vp]
        L.flow2d_host_context.restype = vp
        L.flow2d_host_flow_create.restype = vp
        L.flow2d_host_flow_create.argtypes = [sz, sz, i, i, i]
        L.flow2d_host_flow_destroy.argtypes = [vp]
        L.flow2d_host_flow_pitch.restype = sz
        L.flow2d_host_flow_pitch.argtypes = [vp]
        L.flow2d_host_max_warp_level.restype = sz
        L.flow2d_host_max_warp_level.argtypes = [vp, sz, sz, f]
        L.flow2d_host_compute_flow.argtypes = [vp, fp, fp, fp, fp, C.POINTER(HostParams), fp]
        L.flow2d_host_compute_flow_device.argtypes = [vp, vp, vp, vp, vp, C.POINTER(HostParams), i]
        L.flow2d_host_compute_flow_sequence_device.argtypes = [vp, C.POINTER(vp), sz, C.POINTER(vp), C.POINTER(vp),
                                                               C.POINTER(HostParams)]
        L.flow2d_host_level_timings.restype = sz
        L.flow2d_host_level_timings.argtypes = [vp, fp, sz]
        L.flow2d_host_reset_timings.argtypes = [vp]
        L.flow2d_host_use_graph.argtypes = [vp, i]
        L.flow2d_host_missing_key_leaves_outputs.argtypes = [vp, C.c_char_p]
        L.flow2d_host_read_raw.argtypes = [C.c_char_p, sz, sz, i, fp]
        L.flow2d_host_write_outputs.argtypes = [fp, fp, sz, sz, C.c_char_p, C.c_char_p, f]
        L.flow2d_host_write_raw.argtypes = [fp, sz, sz, i, C.c_char_p]
        L.flow2d_host_max_warp_level_static.restype = sz
        L.flow2d_host_max_warp_level_static.argtypes = [sz, sz, f]
        L.flow2d_host_convert_to_rgb.argtypes = [f, f, C.POINTER(i)]
        L.flow2d_host_load_settings.argtypes = [C.c_char_p, C.POINTER(HostSettings)]
        L.flow2d_host_operator_create.restype = vp
        L.flow2d_host_operator_create.argtypes = [C.c_char_p, sz, sz, sz, i, i]
        L.flow2d_host_operator_name.restype = C.c_char_p
        L.flow2d_host_operator_name.argtypes = [vp]
        L.flow2d_host_operator_execute.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(vp), sz]
        L.flow2d_host_operator_destroy.argtypes = [vp]
        L.flow2d_host_batch_create.restype = vp
        L.flow2d_host_batch_create.argtypes = [sz, sz, i, sz, i, sz]
        L.flow2d_host_batch_group_stride.restype = sz
        L.flow2d_host_batch_group_stride.argtypes = [vp]
        L.flow2d_host_batch_destroy.argtypes = [vp]
        L.flow2d_host_batch_pitch.restype = sz
        L.flow2d_host_batch_pitch.argtypes = [vp]
        L.flow2d_host_batch_lanes.restype = sz
        L.flow2d_host_batch_lanes.argtypes = [vp]
        L.flow2d_host_batch_lane_context.restype = vp
        L.flow2d_host_batch_lane_context.argtypes = [vp, sz]
        L.flow2d_host_batch_use_graph.argtypes = [vp, i]
        L.flow2d_host_batch_compute.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                C.POINTER(HostParams), sz]
        L.flow2d_host_batch_synchronize.argtypes = [vp]
        L.flow2d_host_batch_compute_grouped.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                        C.POINTER(HostParams), sz]
        L.flow2d_host_data2d_create.restype = vp
        L.flow2d_host_data2d_create.argtypes = [sz, sz, i]
        L.flow2d_host_data2d_destroy.argtypes = [vp]
        L.flow2d_host_use_pinned_memory.argtypes = [i]
        L.flow2d_host_data2d_ptr.restype = vp
        L.flow2d_host_data2d_ptr.argtypes = [vp]
        L.flow2d_host_data2d_is_pinned.argtypes = [vp]
        L.flow2d_host_batch_compute_host.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                     C.POINTER(HostParams), sz]
        _host = L
    return _host


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OpticalFlow:
    """OpticalFlow2D of the host layer (Initialize / ComputeFlow / ComputeFlowDevice / Destroy)."""

    def __init__(self, width, height, constancy=GREY, device=0, ctx=None, silent=True, lone=True):
        L = host_lib()
        self._adopted = ctx is not None
        if ctx is not None:
            L.flow2d_host_adopt_context(ctx.handle)
        elif L.flow2d_host_init_device(device) != 0:
            raise Flow2DError(2, "InitDeviceContext")
        self.width, self.height = width, height
        self.handle = L.flow2d_host_flow_create(width, height, _HOST_CONSTANCY[constancy], int(silent), int(bool(lone)))
        if not self.handle:
            if self._adopted:
                L.flow2d_host_adopt_context(None)
            raise Flow2DError(1, "OpticalFlow2D::Initialize")
        self.pitch = L.flow2d_host_flow_pitch(self.handle)

    @staticmethod
    def params(levels, scale, outer, inner, alpha, e_smooth, e_data, median_radius, sigma, algorithm=SOLVER_AUTO,
               sor_omega=0.0):
        return HostParams(levels, scale, outer, inner, alpha, e_smooth, e_data, median_radius, sigma, algorithm,
                          sor_omega)

    def max_warp_level(self, width, height, scale):
        return host_lib().flow2d_host_max_warp_level(self.handle, width, height, scale)

    def compute_flow(self, frame_0, frame_1, params):
        """Host images in, host flow out (upload + pyramid + download).  Returns (u, v, device_ms)."""
        f0 = np.ascontiguousarray(frame_0, np.float32)
        f1 = np.ascontiguousarray(frame_1, np.float32)
        assert f0.shape == (self.height, self.width) and f1.shape == f0.shape
        u = np.empty_like(f0)
        v = np.empty_like(f0)
        ms = C.c_float()
        rc = host_lib().flow2d_host_compute_flow(self.handle, _fptr(f0), _fptr(f1), _fptr(u), _fptr(v),
                                                 C.byref(params), C.byref(ms))
        if rc:
            raise Flow2DError(rc, "OpticalFlow2D::ComputeFlow")
        return u, v, ms.value

    def compute_flow_device(self, dev_f0, dev_f1, dev_u, dev_v, params, timing_mode=0):
        """Device-resident pair (raw device addresses of pitched containers); queued, not synchronised.
        timing_mode: 0 off, 1 events around each level's solve, 2 also around each solver-kernel launch."""
        rc = host_lib().flow2d_host_compute_flow_device(self.handle, dev_f0, dev_f1, dev_u, dev_v, C.byref(params),
                                                        int(timing_mode))
        if rc:
            raise Flow2DError(rc, "OpticalFlow2D::ComputeFlowDevice")

    def compute_flow_sequence_device(self, dev_frames, dev_us, dev_vs, params):
        """Flows of an image sequence: flow k goes from dev_frames[k] to dev_frames[k + 1] into (dev_us[k], dev_vs[k]).
        Every frame's blurred plane and pyramid levels are computed once.  Queued, not synchronised."""
        n = len(dev_frames)
        if n < 2 or len(dev_us) != n - 1 or len(dev_vs) != n - 1:
            raise ValueError("a sequence of n frames takes n - 1 flow plane pairs")
        frames = (C.c_void_p * n)(*dev_frames)
        us = (C.c_void_p * (n - 1))(*dev_us)
        vs = (C.c_void_p * (n - 1))(*dev_vs)
        rc = host_lib().flow2d_host_compute_flow_sequence_device(self.handle, frames, n, us, vs, C.byref(params))
        if rc:
            raise Flow2DError(rc, "OpticalFlow2D::ComputeFlowSequenceDevice")

    def level_timings(self):
        """[(width, height, solve_ms, kernel_ms, kernel_launches, algorithmic_bytes_per_launch, algorithm)] per level;
        algorithm = the flow2d_solver_algorithm the level actually ran (never AUTO)."""
        cap = 8192
        buf = np.zeros(7 * cap, np.float32)
        n = host_lib().flow2d_host_level_timings(self.handle, _fptr(buf), cap)
        return [(int(buf[7 * i]), int(buf[7 * i + 1]), float(buf[7 * i + 2]), float(buf[7 * i + 3]),
                 int(buf[7 * i + 4]), float(buf[7 * i + 5]), int(buf[7 * i + 6])) for i in range(min(n, cap))]

    def use_graph(self, on=True):
        """Record the pyramid of a (buffers, parameters) combination once and replay it (HIP graph)."""
        host_lib().flow2d_host_use_graph(self.handle, int(on))

    def reset_timings(self):
        host_lib().flow2d_host_reset_timings(self.handle)

    def missing_key_leaves_outputs(self, key):
        return host_lib().flow2d_host_missing_key_leaves_outputs(self.handle, key.encode())

    def close(self):
        if self.handle:
            host_lib().flow2d_host_flow_destroy(self.handle)
            self.handle = None
            if self._adopted:  # the caller owns (and may now destroy) the adopted context
                host_lib().flow2d_host_adopt_context(None)


class OpticalFlowBatch:
    """OpticalFlowBatch2D of the host layer: independent pairs spread over `lanes` (stream + OpticalFlow2D + plane
    pool each) on one GPU; pair k of a call runs on lane (first_lane + k) mod lanes.  The scheduling is C++; this
    class only marshals device addresses."""

    def __init__(self, width, height, constancy=GREY, lanes=4, device=0, group_size=1):
        L = host_lib()
        self.width, self.height, self.group_size = width, height, group_size
        self.handle = L.flow2d_host_batch_create(width, height, _HOST_CONSTANCY[constancy], lanes, device, group_size)
        if not self.handle:
            raise Flow2DError(1, "OpticalFlowBatch2D::Initialize")
        self.pitch = L.flow2d_host_batch_pitch(self.handle)
        self.lanes = L.flow2d_host_batch_lanes(self.handle)
        # group_size > 1: every plane handed to compute_flow_batch_device is a tall container, pair g of the group
        # group_stride bytes * g behind its address (= pitch * height: the pairs' containers one below the other)
        self.group_stride = L.flow2d_host_batch_group_stride(self.handle)

    params = staticmethod(OpticalFlow.params)

    def use_graph(self, on=True):
        host_lib().flow2d_host_batch_use_graph(self.handle, int(on))

    def compute_flow_batch_device(self, dev_f0s, dev_f1s, dev_us, dev_vs, params, first_lane=0):
        """Raw device addresses of pitched containers, one entry per pair; queued, not synchronised."""
        n = len(dev_f0s)
        if not (len(dev_f1s) == len(dev_us) == len(dev_vs) == n):
            raise ValueError("one frame 0, frame 1, u and v plane per pair")
        arrays = [(C.c_void_p * n)(*a) for a in (dev_f0s, dev_f1s, dev_us, dev_vs)]
        rc = host_lib().flow2d_host_batch_compute(self.handle, n, *arrays, C.byref(params), first_lane)
        if rc:
            raise Flow2DError(rc, "OpticalFlowBatch2D::ComputeFlowBatchDevice")

    def compute_flow_batch_device_grouped(self, dev_f0s, dev_f1s, dev_us, dev_vs, params, first_lane=0):
        """Independent pairs (one container per plane); the object forms the lock-step groups (gather, group, hand back)."""
        n = len(dev_f0s)
        if not (len(dev_f1s) == len(dev_us) == len(dev_vs) == n):
            raise ValueError("one frame 0, frame 1, u and v plane per pair")
        arrays = [(C.c_void_p * n)(*a) for a in (dev_f0s, dev_f1s, dev_us, dev_vs)]
        rc = host_lib().flow2d_host_batch_compute_grouped(self.handle, n, *arrays, C.byref(params), first_lane)
        if rc:
            raise Flow2DError(rc, "OpticalFlowBatch2D::ComputeFlowBatchDeviceGrouped")

    def compute_flow_batch(self, frames_0, frames_1, flows_u, flows_v, params, first_lane=0):
        """OpticalFlowBatch2D::ComputeFlowBatch: HostImage objects in and out, uploads / downloads pipelined against the
        lanes' pyramids.  Queued: read the flows after synchronize()."""
        n = len(frames_0)
        if not (len(frames_1) == len(flows_u) == len(flows_v) == n):
            raise ValueError("one frame 0, frame 1, u and v image per pair")
        arrays = [(C.c_void_p * n)(*[q.handle for q in a]) for a in (frames_0, frames_1, flows_u, flows_v)]
        rc = host_lib().flow2d_host_batch_compute_host(self.handle, n, *arrays, C.byref(params), first_lane)
        if rc:
            raise Flow2DError(rc, "OpticalFlowBatch2D::ComputeFlowBatch")

    def synchronize(self):
        if host_lib().flow2d_host_batch_synchronize(self.handle) != 0:
            raise Flow2DError(2, "OpticalFlowBatch2D::Synchronize")

    def close(self):
        if self.handle:
            host_lib().flow2d_host_batch_destroy(self.handle)
            self.handle = None


class HostImage:
    """A Data2D of the host layer (tight row-major float32), in page-locked memory when pinned=True
    (HostMemory::Pinned).  `array` is a numpy view of its pixels, valid until close()."""

    def __init__(self, width, height, pinned=True, data=None):
        L = host_lib()
        self.handle = L.flow2d_host_data2d_create(width, height, int(pinned))
        if not self.handle:
            raise MemoryError("Data2D(%d, %d)" % (width, height))
        self.width, self.height = width, height
        self.pinned = bool(L.flow2d_host_data2d_is_pinned(self.handle))
        buf = (C.c_float * (width * height)).from_address(L.flow2d_host_data2d_ptr(self.handle))
        self.array = np.frombuffer(buf, np.float32).reshape(height, width)
        if data is not None:
            self.array[...] = data

    def close(self):
        if self.handle:
            self.array = None
            host_lib().flow2d_host_data2d_destroy(self.handle)
            self.handle = None


def read_raw(path, width, height, u8):
    out = np.empty((height, width), np.float32)
    rc = host_lib().flow2d_host_read_raw(path.encode(), width, height, int(u8), _fptr(out))
    return out if rc == 0 else None


def write_raw(image, path, u8):
    a = np.ascontiguousarray(image, np.float32)
    return host_lib().flow2d_host_write_raw(_fptr(a), a.shape[1], a.shape[0], int(u8), path.encode()) == 0


def max_warp_level(width, height, scale):
    """OpticalFlowBase2D::GetMaxWarpLevel of the host layer (no device needed)."""
    return host_lib().flow2d_host_max_warp_level_static(width, height, scale)


def write_outputs(u, v, ppm_path, amp_path, flow_max_scale=10.0):
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    host_lib().flow2d_host_write_outputs(_fptr(u), _fptr(v), u.shape[1], u.shape[0], ppm_path.encode(),
                                         amp_path.encode(), flow_max_scale)


def convert_to_rgb(x, y):
    rgb = (C.c_int * 3)()
    host_lib().flow2d_host_convert_to_rgb(x, y, rgb)
    return tuple(rgb)


def load_settings(path):
    s = HostSettings()
    rc = host_lib().flow2d_host_load_settings(path.encode(), C.byref(s))
    return s if rc == 0 else None


class Operator:
    """One of the reference's six operator classes (CudaOperation*2D) behind its Initialize/Execute bag API.

    execute(**bag): every value is a ctypes object (c_ulonglong device pointer, c_size_t, c_float, DataSize3...)
    whose address is pushed under its keyword, exactly like OperationParameters::PushValuePtr."""

    def __init__(self, kind, container_width, container_height, pitch_bytes, constancy=GREY, ctx=None,
                 omit_container_size=False):
        L = host_lib()
        if ctx is not None:
            L.flow2d_host_adopt_context(ctx.handle)
        self.handle = L.flow2d_host_operator_create(kind.encode(), container_width, container_height, pitch_bytes,
                                                    constancy, int(omit_container_size))
        if not self.handle:
            raise Flow2DError(1, "CudaOperation%s2D::Initialize" % kind.capitalize())

    @property
    def name(self):
        return host_lib().flow2d_host_operator_name(self.handle).decode()

    def execute(self, **bag):
        keys = (C.c_char_p * len(bag))(*[k.encode() for k in bag])
        vals = (C.c_void_p * len(bag))(*[C.cast(C.pointer(v), C.c_void_p) for v in bag.values()])
        host_lib().flow2d_host_operator_execute(self.handle, keys, vals, len(bag))

    def close(self):
        if self.handle:
            host_lib().flow2d_host_operator_destroy(self.handle)
            self.handle = None


class DataSize3(C.Structure):
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t), ("pitch", C.c_size_t)]
