"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol the
header declares, the oracle still reproduces the committed golden vectors, and the host-side
logic that does not touch the device (settings reader, raw IO, colour wheel) behaves like the
reference's."""
import ctypes
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "flow2d_golden.npz")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()


def test_c_abi_exports_every_declared_symbol(flow2d):
    header = open(os.path.join(ROOT, "include", "flow2d_c_abi.h")).read()
    declared = set(re.findall(r"FLOW2D_API\s+[\w\s\*]+?\b(flow2d_\w+)\s*\(", header))
    assert len(declared) >= 35
    lib = flow2d.hip_lib()
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, "declared in include/flow2d_c_abi.h but not exported: %s" % missing
    assert lib.flow2d_abi_version() == 1
    assert lib.flow2d_status_string(5) == b"unsupported parameter"
    # no torch / C++ types in the signatures: the header must compile as plain C
    src = os.path.join(ROOT, "include", "flow2d_c_abi.h")
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", src])


def test_c_abi_rejects_bad_arguments_without_a_device(flow2d):
    lib = flow2d.hip_lib()
    assert lib.flow2d_context_destroy(None) == 1
    assert lib.flow2d_plane_pitch_bytes(584) == 2560 and lib.flow2d_plane_pitch_bytes(64) == 256
    taps = (ctypes.c_float * 51)()
    r = ctypes.c_int()
    assert lib.flow2d_gaussian_kernel(-1.0, taps, ctypes.byref(r)) == 1
    assert lib.flow2d_gaussian_kernel(9.0, taps, ctypes.byref(r)) == 5  # 55 taps > 51


def test_solver_algorithm_selection_and_32bit_guard(flow2d):
    """flow2d_solve_level's choice of algorithm, checked without a device: AUTO = LDS tiles up to 600 x 600 pixels (one workgroup
    up to 64 x 32 where the tiled kernel does not apply), the fused strip kernel above (when there are >= 2 sweeps to fuse), per-sweep launches
    otherwise -- and whenever the plane reaches
    4 GiB, which the fused kernel's 32-bit buffer offsets cannot address (an explicit FUSED request is refused there
    instead of wrapping around)."""
    raw = flow2d.hip_lib().flow2d_solver_algorithm_for
    pick = lambda req, w, h, pitch_bytes, outer, inner, constancy=0: raw(req, w, h, pitch_bytes, outer, inner, constancy)
    AUTO, SWEEP, FUSED, ONE, TILED = 0, 1, 2, 3, 4
    pitch = lambda w: flow2d.hip_lib().flow2d_plane_pitch_bytes(w)
    assert pick(AUTO, 64, 32, pitch(64), 10, 5) == TILED and pick(AUTO, 64, 33, pitch(64), 10, 5) == TILED
    assert pick(AUTO, 64, 32, pitch(64), 10, 5, flow2d.LOG_DERIVATIVES) == ONE and pick(AUTO, 64, 32, pitch(64), 10, 1) == ONE
    assert pick(AUTO, 512, 512, pitch(512), 10, 5) == TILED and pick(AUTO, 512, 512, pitch(512), 10, 7) == FUSED
    assert pick(AUTO, 600, 600, pitch(600), 10, 5) == TILED and pick(AUTO, 601, 600, pitch(601), 10, 5) == FUSED
    assert pick(AUTO, 600, 600, pitch(600), 10, 5, flow2d.GRADIENT) == TILED and pick(AUTO, 640, 640, pitch(640), 10, 5, flow2d.GRADIENT) == FUSED
    assert pick(AUTO, 1024, 1024, pitch(1024), 10, 5) == FUSED and pick(AUTO, 1024, 1024, pitch(1024), 10, 5, flow2d.GRADIENT) == FUSED
    assert pick(AUTO, 1920, 1080, pitch(1920), 10, 5, flow2d.GRADIENT) == FUSED
    assert pick(AUTO, 512, 512, pitch(512), 10, 5, flow2d.LOG_DERIVATIVES) == FUSED  # no tiled Log kernel
    assert pick(TILED, 1024, 1024, pitch(1024), 10, 6) == -1 and pick(TILED, 4096, 4096, pitch(4096), 10, 5) == TILED
    assert pick(AUTO, 4096, 4096, pitch(4096), 10, 5) == FUSED and pick(AUTO, 4096, 4096, pitch(4096), 10, 1) == SWEEP
    assert pick(ONE, 65, 40, pitch(65), 1, 1) == -1 and pick(FUSED, 640, 480, pitch(640), 2, 0) == -1
    assert pick(FUSED, 640, 480, pitch(640), 0, 0) == FUSED and pick(7, 64, 64, 256, 1, 1) == -1
    # 32768 x 32767 floats: 4 GiB minus one row -> still addressable; 32768 x 32768: exactly 4 GiB -> not
    assert pick(FUSED, 32768, 32767, pitch(32768), 10, 5) == FUSED and pick(AUTO, 32768, 32767, pitch(32768), 10, 5) == FUSED
    assert pick(FUSED, 32768, 32768, pitch(32768), 10, 5) == -1 and pick(AUTO, 32768, 32768, pitch(32768), 10, 5) == SWEEP
    assert pick(AUTO, 100000, 20000, pitch(100000), 10, 5) == SWEEP


def test_gaussian_taps_match_oracle(flow2d, oracle):
    for sigma in (0.45, 1.0, 1.5, 3.0, 8.3):
        taps, r = flow2d.gaussian_kernel(sigma)
        otaps, orad = oracle.gaussian_taps(sigma)
        assert r == orad and np.array_equal(taps, otaps)


def test_oracle_reproduces_golden_vectors(oracle):
    g = np.load(GOLDEN)
    f0, f1 = g["small_f0"], g["small_f1"]
    for name, constancy in (("grey", 0), ("grad", 1)):
        stages = {}
        u, v, _ = oracle.compute_flow(f0, f1, 6, 0.8, 2, 3, 3.5, 0.001, 0.001, 5, 0.45, constancy,
                                      dump=lambda t, l, p: stages.__setitem__("%s_L%d" % (t, l), p.copy()))
        assert np.array_equal(u, g["small_%s_u" % name]) and np.array_equal(v, g["small_%s_v" % name])
        for k in ("blur0_L-1", "frame0_res_L3", "warped_L2", "phi_L1", "ksi_L1", "du_L1", "dv_L1", "flow_u_add_L1",
                  "flow_u_med_L1", "flow_u_res_L0"):
            assert np.array_equal(stages[k], g["small_%s_%s" % (name, k)]), k
    for s in (0.45, 1.5):
        assert np.array_equal(oracle.gaussian_taps(s)[0], g["taps_%g" % s])
    for w, h, s100, want in g["level_table"]:
        assert oracle.max_warp_level(int(w), int(h), s100 / 100.0) == want


def test_oracle_reproduces_rub_golden(oracle):
    from test_oracle import rub_pair
    g = np.load(GOLDEN)
    r1, r2 = rub_pair()
    u, v, _ = oracle.compute_flow(r1, r2, 8, 0.8, 3, 5, 3.5, 0.001, 0.001, 5, 0.45)
    assert [sha(u), sha(v)] == list(g["rub_short_sha"])
    assert np.array_equal(u[::2, ::2], g["rub_short_u_sub2"])


def test_settings_reader(flow2d, tmp_path):
    xml = """<?xml version="1.0"?>
<!-- Settings for the Optical flow computation program -->
<OpticalFlow>
  <Input>
    <Path inputPath="./data/"/>
    <Mode Nx="584" Ny="388" imageType="8-bit">
    	<Files file1 ="rub1.raw" file2 ="rub2.raw"/>
    </Mode>
  </Input>
  <Parameters>
    <Method mode ="2d" run="flow" key="0" />
    <Solver>
      <Iterations inner="5" outer="20"/>
      <Warping levels="20" scaling="0.9" medianRadius="5"/>
      <Model sigma="0.45" alpha ="3.5" e_smooth="0.001" e_data="0.001"/>
    </Solver>
  </Parameters>
  <Output>
    <Path outputPath="./out/"/>
  </Output>
</OpticalFlow>
"""
    p = tmp_path / "settings.xml"
    p.write_text(xml)
    s = flow2d.load_settings(str(p))
    assert s is not None
    assert (s.width, s.height, s.iterInner, s.iterOuter, s.levels, s.medianRadius) == (584, 388, 5, 20, 20, 5)
    assert abs(s.sigma - 0.45) < 1e-7 and abs(s.alpha - 3.5) < 1e-7 and abs(s.warpScale - 0.9) < 1e-7
    assert abs(s.e_smooth - 0.001) < 1e-9 and abs(s.e_data - 0.001) < 1e-9
    assert s.inputPath == b"./data/" and s.outputPath == b"./out/"
    assert s.fileName1 == b"rub1.raw" and s.fileName2 == b"rub2.raw"
    assert s.imageType == b"8-bit" and s.dataConstancy == b"grey" and s.press_key == 0
    # a missing element is an error (-1 -> exit code 3 in the CLI), not a crash as in the reference
    q = tmp_path / "broken.xml"
    q.write_text(xml.replace('<Warping levels="20" scaling="0.9" medianRadius="5"/>', ""))
    assert flow2d.load_settings(str(q)) is None
    assert flow2d.load_settings(str(tmp_path / "absent.xml")) is None


def test_shipped_rub_settings_file(flow2d):
    s = flow2d.load_settings(os.path.join(ROOT, "cuda-flow2d_amd", "host", "settings_rub.xml"))
    assert s is not None and (s.width, s.height) == (584, 388) and s.imageType == b"8-bit"


def test_raw_readers(flow2d, tmp_path):
    rng = np.random.default_rng(5)
    a8 = rng.integers(0, 256, (7, 9), dtype=np.uint8)
    a32 = rng.normal(0, 1, (7, 9)).astype(np.float32)
    (tmp_path / "a8.raw").write_bytes(a8.tobytes())
    (tmp_path / "a32.raw").write_bytes(a32.tobytes())
    assert np.array_equal(flow2d.read_raw(str(tmp_path / "a8.raw"), 9, 7, True), a8.astype(np.float32))
    assert np.array_equal(flow2d.read_raw(str(tmp_path / "a32.raw"), 9, 7, False), a32)
    # wrong dimensions (too short / trailing bytes) and missing files are rejected, no crash
    assert flow2d.read_raw(str(tmp_path / "a8.raw"), 9, 8, True) is None
    assert flow2d.read_raw(str(tmp_path / "a8.raw"), 9, 6, True) is None
    assert flow2d.read_raw(str(tmp_path / "a8.raw"), 9, 7, False) is None
    assert flow2d.read_raw(str(tmp_path / "nope.raw"), 9, 7, True) is None


_libm = ctypes.CDLL("libm.so.6")
_libm.atanf.restype = ctypes.c_float
_libm.atanf.argtypes = [ctypes.c_float]


def _atanf(t):
    # the reference calls the float overload of atan (io_utils.cpp:24 includes <math.h> in C++);
    # numpy's own float32 arctan can differ from libm's in the last bit
    return np.float32(_libm.atanf(float(t)))


def _wheel_reference(x, y):
    """The Bruhn colour wheel as io_utils.cpp:140-225 words it, in numpy scalars."""
    f = np.float32
    pi = f(2.0 * np.arccos(0.0))
    x, y = f(x), f(y)
    amp = np.sqrt(x * x + y * y)
    amp = f(1) if amp > 1 else amp
    if x == 0:
        phi = f(0.5 * float(pi)) if y >= 0 else f(1.5 * float(pi))
    elif x > 0:
        phi = _atanf(y / x) if y >= 0 else f(2.0 * float(pi) + float(_atanf(y / x)))
    else:
        phi = f(float(pi) + float(_atanf(y / x)))
    phi = f(float(phi) / 2.0)
    P = float(pi)
    segs = [(0.0, 0.125, (255, 0, 0), (255, 0, 255)), (0.125, 0.25, (255, 0, 255), (64, 64, 255)),
            (0.25, 0.375, (64, 64, 255), (0, 255, 255)), (0.375, 0.5, (0, 255, 255), (0, 255, 0)),
            (0.5, 0.75, (0, 255, 0), (255, 255, 0)), (0.75, 1.0, (255, 255, 0), (255, 0, 0))]
    rgb = [0, 0, 0]
    for lo, hi, c0, c1 in segs:
        inside = (float(phi) >= lo * P) and (float(phi) <= hi * P if hi == 1.0 else float(phi) < hi * P)
        if inside:
            beta = f((float(phi) - lo * P) / ((hi - lo) * P))
            alpha = f(1.0 - float(beta))
            rgb = [int(np.floor(float(amp) * (float(alpha) * a + float(beta) * b))) for a, b in zip(c0, c1)]
    return tuple(255 if c >= 255 else (c if c > 0 else 0) for c in rgb)


def test_colour_wheel_and_output_files(flow2d, tmp_path):
    rng = np.random.default_rng(9)
    pts = [(0, 0), (1, 0), (0, 1), (-1, 0), (0, -1), (0.3, 0.3), (-2, 5), (0.5, -0.1), (1e-3, -1e-3)]
    pts += [tuple(p) for p in rng.uniform(-1.5, 1.5, (300, 2))]
    for x, y in pts:
        assert flow2d.convert_to_rgb(x, y) == _wheel_reference(x, y), (x, y)
    u = rng.normal(0, 4, (6, 5)).astype(np.float32)
    v = rng.normal(0, 4, (6, 5)).astype(np.float32)
    ppm, amp = str(tmp_path / "res.pgm"), str(tmp_path / "amp.raw")
    flow2d.write_outputs(u, v, ppm, amp, 10.0)
    data = open(ppm, "rb").read()
    header = b"P6 \n5 6 \n255\n"
    assert data.startswith(header) and len(data) == len(header) + 5 * 6 * 3
    px = np.frombuffer(data[len(header):], np.uint8).reshape(6, 5, 3)
    k = np.float32(1.0 / 10.0)
    for yy in range(6):
        for xx in range(5):
            assert tuple(px[yy, xx]) == _wheel_reference(u[yy, xx] * k, v[yy, xx] * k)
    mag = np.fromfile(amp, np.float32).reshape(6, 5)
    assert np.array_equal(mag, np.sqrt(u * u + v * v))


def test_cli_exit_codes_without_device(flow2d, tmp_path):
    """Exit code 1 = no device (main.cpp:51-54); only meaningful where no GPU is visible."""
    if flow2d.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device exit path cannot be exercised")
    rc = subprocess.call([flow2d.CLI_PATH, "nope.xml"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert rc == 1


def test_committed_bench_line_has_the_contract_fields():
    """The bench line committed under profiles/ carries every field the bench contract names.  The roofline block quotes
    the PHYSICAL HBM fraction of the dominant kernel (PMC bytes over the launch time) as frac, the contract's algorithmic
    figure as effective_frac, and names what the kernel is bound by."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.loads(open(os.path.join(root, "profiles", "r03_default_bench_line.json")).read())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "output_check", "batch",
                "reference_gpu_baseline", "pairs_per_s", "pairs_per_s_incl_h2d", "host_entry"):
        assert key in line, key
    assert line["config"]["workload"] == "cfg3_4096_gradient" and line["scaling"] == "weak" and line["dtype"] == "f32"
    assert line["output_check"]["ok"] and line["batch"]["output_check"]["ok"]
    assert line["host_entry"]["flows_bit_identical_to_device_resident_run"]
    assert 0 < line["pairs_per_s_incl_h2d"] <= line["pairs_per_s"]      # the H<->D-inclusive bracket cannot be faster
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "effective_achieved", "effective_frac",
                "algorithmic_bytes_per_launch", "valu_instr_per_launch", "valu_issue_frac", "pmc_source"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "valu") and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert abs(roof["achieved"] - roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    assert abs(roof["effective_frac"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-3
    # the fused kernel moves a fraction of the algorithmic bytes: physically far below the HBM peak, bound by VALU issue
    assert roof["bound"] == "valu" and 0.05 < roof["frac"] < 0.6 < roof["effective_frac"]
    assert roof["pmc_source"].startswith("this run")                    # counters measured in the run, not quoted


# ---- the product's host layer against THE REFERENCE'S OWN host code ---------------------------------------
# tests/golden/ref_host_golden.npz was produced by running the reference's host sources (compiled where they lie)
# in the build container: tests/golden/make_ref_host_golden.py.
@pytest.fixture(scope="module")
def ref_host_golden():
    import json
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_host_golden.npz"))
    return g, json.loads(str(g["meta"]))


def test_ref_host_levels_and_taps(flow2d, ref_host_golden):
    _, meta = ref_host_golden
    for w, h, s, want in meta["levels"]:
        assert flow2d.max_warp_level(w, h, s) == want, (w, h, s)
    for sigma, rec in meta["taps"].items():
        taps, r = flow2d.gaussian_kernel(float(sigma))
        assert r == rec["radius"] and ["%08x" % b for b in taps.view(np.uint32)] == rec["bits"], sigma


def test_ref_host_parameter_bag(flow2d, ref_host_golden):
    _, meta = ref_host_golden
    # first push accepted, second refused, value of the first kept, missing key -> null, Clear() empties
    assert meta["bag"] == [1, 0, 1, 0, 0]
    assert flow2d.host_lib().flow2d_host_bag_selftest() == 0  # asserts exactly those five facts


def test_ref_host_settings(flow2d, ref_host_golden, tmp_path):
    _, meta = ref_host_golden
    bits = lambda x: "%08x" % np.float32(x).view(np.uint32)
    p = tmp_path / "variant.xml"
    p.write_text(meta["settings_variant_xml"])
    ref = meta["settings_variant"]
    s = flow2d.load_settings(str(p))
    assert ref["rc"] == "0" and s is not None
    assert [s.width, s.height, s.iterInner, s.iterOuter, s.levels, s.medianRadius] == \
        [int(ref[k]) for k in ("width", "height", "inner", "outer", "levels", "medianRadius")]
    assert [bits(s.sigma), bits(s.alpha), bits(s.e_smooth), bits(s.e_data), bits(s.warpScale)] == \
        [ref[k] for k in ("sigma", "alpha", "e_smooth", "e_data", "scaling")]
    assert [s.inputPath.decode(), s.outputPath.decode(), s.fileName1.decode(), s.fileName2.decode()] == \
        [ref[k] for k in ("inputPath", "outputPath", "file1", "file2")]
    assert meta["settings_missing_file"]["rc"] == "-1"
    # the reference's shipped settings.xml, field by field as the reference itself parses it
    ref = meta["settings_reference_file"]
    assert (ref["width"], ref["height"], ref["inner"], ref["outer"], ref["levels"], ref["medianRadius"]) == \
        ("128", "128", "5", "20", "20", "5")
    assert (ref["sigma"], ref["alpha"], ref["e_smooth"], ref["scaling"]) == \
        (bits(0.45), bits(3.5), bits(0.001), bits(0.9))


def test_ref_host_raw_io(flow2d, ref_host_golden, tmp_path):
    g, _ = ref_host_golden
    (tmp_path / "a8.raw").write_bytes(g["raw_u8_in"].tobytes())
    h, w = g["raw_u8_in"].shape
    assert np.array_equal(flow2d.read_raw(str(tmp_path / "a8.raw"), w, h, True), g["raw_u8_as_f32"])
    assert flow2d.write_raw(g["raw_f32_in"], str(tmp_path / "f8.raw"), True)
    assert np.array_equal(np.fromfile(tmp_path / "f8.raw", np.uint8).reshape(h, w), g["raw_f32_as_u8"])


def test_ref_host_colour_wheel_ppm_and_magnitude(flow2d, ref_host_golden, tmp_path):
    """WriteFlowToImageRGB / WriteMagnitudeToFileF32 (io_utils.cpp:35-114,140-225): byte-identical files."""
    g, _ = ref_host_golden
    ppm, amp = str(tmp_path / "res.pgm"), str(tmp_path / "amp.raw")
    flow2d.write_outputs(g["flow_u"], g["flow_v"], ppm, amp, 10.0)
    assert open(ppm, "rb").read() == g["ppm_bytes"].tobytes()
    assert np.array_equal(np.fromfile(amp, np.float32).reshape(g["amp"].shape).view(np.uint32), g["amp"].view(np.uint32))


def test_host_library_exports_the_batch_entry(flow2d):
    """The C++ host layer carries the batched path (OpticalFlowBatch2D) and the C-ABI the lock-step batch switch;
    a batch of zero lanes is refused without touching a device."""
    host = flow2d.host_lib()
    for name in ("flow2d_host_batch_create", "flow2d_host_batch_compute", "flow2d_host_batch_synchronize",
                 "flow2d_host_batch_destroy", "flow2d_host_batch_group_stride", "flow2d_host_batch_use_graph"):
        assert hasattr(host, name), name
    assert hasattr(flow2d.hip_lib(), "flow2d_context_set_batch")
    assert flow2d.hip_lib().flow2d_context_set_batch(None, 2, 4096) == 1  # no context: invalid argument
    assert not host.flow2d_host_batch_create(64, 64, 0, 0, 0, 1)         # zero lanes


def test_plain_c_client_links_and_runs_without_a_device(flow2d, tmp_path):
    """tests/c/abi_link.c: a C99 program against include/flow2d_c_abi.h and libflow2d_hip.so (no C++ in between)."""
    exe = tmp_path / "abi_link"
    csrc = os.path.join(ROOT, "cuda-flow2d_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_link.c"), "-L", csrc, "-lflow2d_hip",
                           "-Wl,-rpath," + csrc, "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "flow2d C-ABI v1" in out.stdout


def test_hardware_queues_are_requested_not_set_at_load(flow2d):
    """Loading libflow2d_hip.so leaves the environment alone (rounds 2-5 set GPU_MAX_HW_QUEUES from a library constructor);
    flow2d_request_hw_queues(8) sets it when the caller has not and HIP is not running yet, leaves a caller's value alone and
    says so, and refuses once the process holds the driver's device node (a stand-in: any open descriptor of /dev/kfd)."""
    code = ("import ctypes, os; L = ctypes.CDLL(%r); L.flow2d_last_error.restype = ctypes.c_char_p; a = L.flow2d_hw_queues();"
            "rc = L.flow2d_request_hw_queues(8); print(a, rc, L.flow2d_hw_queues(), L.flow2d_last_error().decode()[:40].replace(' ', '_') or '-')"
            % flow2d.HIP_LIB_PATH)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out[:3] == ["4", "0", "8"], out  # untouched by the load, granted by the request
    out = subprocess.run([os.sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="2"), capture_output=True,
                         text=True, check=True).stdout.split()
    assert out[:3] == ["2", "5", "2"] and out[3].startswith("GPU_MAX_HW_QUEUES"), out
    out = subprocess.run([os.sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="16"), capture_output=True,
                         text=True, check=True).stdout.split()
    assert out[:3] == ["16", "0", "16"], out
    if os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK):  # (a GPU box, or a container that maps the node)
        late = "import os; fd = os.open('/dev/kfd', os.O_RDONLY); " + code
        out = subprocess.run([os.sys.executable, "-c", late], env=env, capture_output=True, text=True, check=True).stdout.split()
        assert out[:3] == ["4", "5", "4"] and "HIP" in out[3], out
