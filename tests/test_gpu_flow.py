"""End-to-end parity of OpticalFlow2D::ComputeFlow (C++ host layer over the C-ABI, HIP kernels) against
the CPU oracle and the committed golden vectors.  Gate of BASELINE.json: RMSE(u), RMSE(v) <= 1e-4 at
identical iteration counts; the path is built without FMA contraction so the fields are in fact
bit-identical, which is what these tests assert."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from test_oracle import rub_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "flow2d_golden.npz")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


@pytest.fixture()
def make_flow(flow2d):
    made = []

    def _make(w, h, constancy=0):
        assert flow2d.device_count() > 0, "no HIP device: gpu tests need the MI355X box"
        f = flow2d.OpticalFlow(w, h, constancy)
        made.append(f)
        return f

    yield _make
    for f in made:
        f.close()


@pytest.mark.parametrize("algorithm", [0, 1])
def test_rub_settings_xml_parity(flow2d, oracle, make_flow, algorithm):
    """Config 1: rub1 <-> rub2 (u8 -> f32), solver values of the reference's settings.xml."""
    r1, r2 = rub_pair()
    flow = make_flow(584, 388)
    p = flow.params(20, 0.9, 20, 5, 3.5, 0.001, 0.001, 5, 0.45, algorithm)
    u, v, ms = flow.compute_flow(r1, r2, p)
    ou, ov, _ = oracle.compute_flow(r1, r2, 20, 0.9, 20, 5, 3.5, 0.001, 0.001, 5, 0.45)
    assert rmse(u, ou) <= 1e-4 and rmse(v, ov) <= 1e-4      # the BASELINE gate
    assert np.array_equal(u, ou) and np.array_equal(v, ov)  # and in fact bit-identical
    g = np.load(GOLDEN)
    assert [sha(u), sha(v)] == list(g["rub_settings_sha"])
    assert np.array_equal(u[::4, ::4], g["rub_settings_u_sub4"])
    assert np.array_equal(v[::4, ::4], g["rub_settings_v_sub4"])
    # SURVEY 8(c) anchors recorded from the reference's own sources
    assert abs(float(u[194, 292]) - 1.246711) < 1e-6 and abs(float(v[194, 292]) + 1.048284) < 1e-6
    assert ms > 0


def test_rub_main_defaults_parity(flow2d, oracle, make_flow):
    """Secondary parameter set: main.cpp defaults (47 levels, 40 x 5 sweeps, alpha 35, sigma 1.5)."""
    r1, r2 = rub_pair()
    flow = make_flow(584, 388)
    p = flow.params(50, 0.9, 40, 5, 35.0, 0.001, 0.001, 5, 1.5)
    u, v, _ = flow.compute_flow(r1, r2, p)
    ou, ov, _ = oracle.compute_flow(r1, r2, 50, 0.9, 40, 5, 35.0, 0.001, 0.001, 5, 1.5)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)


@pytest.mark.parametrize("constancy,name", [(0, "grey"), (1, "grad")])
def test_small_pair_against_golden(flow2d, make_flow, constancy, name):
    g = np.load(GOLDEN)
    flow = make_flow(100, 70, constancy)
    u, v, _ = flow.compute_flow(g["small_f0"], g["small_f1"], flow.params(6, 0.8, 2, 3, 3.5, 0.001, 0.001, 5, 0.45))
    assert np.array_equal(u, g["small_%s_u" % name]) and np.array_equal(v, g["small_%s_v" % name])


@pytest.mark.parametrize("w,h,levels,scale,outer,inner,median,sigma,constancy", [
    (256, 192, 5, 0.5, 10, 5, 5, 1.5, 0),    # config-2 shaped, small
    (256, 128, 4, 0.5, 3, 5, 5, 1.5, 1),     # gradient constancy, every level a 16x8 multiple
    (256, 128, 4, 0.5, 3, 5, 5, 1.5, 2),     # gradient constancy over true neighbours (opt-in, not in the reference)
    (100, 70, 6, 0.8, 2, 3, 5, 0.45, 2),
    (131, 77, 30, 0.9, 2, 2, 3, 0.0, 0),     # no pre-blur, median 3, deep pyramid down to 4-5 px
    (64, 48, 3, 0.7, 1, 1, 7, 0.45, 0),      # median 7
    (96, 64, 3, 0.6, 2, 3, 1, 0.45, 1),      # median width 1 = copy
    (80, 60, 2, 0.5, 2, 2, 6, 0.45, 0),      # even width -> 5
    (640, 528, 3, 0.5, 2, 7, 5, 1.5, 1),     # 7 sweeps per outer iteration: two fused launches at the finest level
    (1024, 600, 2, 0.5, 1, 10, 5, 1.5, 0),   # 10 sweeps: 5 + 5
])
def test_synthetic_pairs_parity(flow2d, oracle, make_flow, w, h, levels, scale, outer, inner, median, sigma,
                                constancy):
    f0, f1 = oracle.synthetic_pair(w, h, 1.5, -0.75, seed=1, noise=True)
    flow = make_flow(w, h, constancy)
    u, v, _ = flow.compute_flow(f0, f1, flow.params(levels, scale, outer, inner, 35.0, 0.001, 0.001, median, sigma))
    ou, ov, _ = oracle.compute_flow(f0, f1, levels, scale, outer, inner, 35.0, 0.001, 0.001, median, sigma,
                                    constancy)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)


def _random_cases(count, seed):
    """Deterministic spread of pipeline configurations: odd sizes, every scale from shallow to deep pyramids, 1-7 sweeps
    (beyond 5: chunked fused launches), the three median widths, with and without the pre-blur, alpha over two decades."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        w, h = int(rng.integers(9, 420)), int(rng.integers(9, 300))
        constancy = int(rng.choice([0, 0, 1, 2]))
        cases.append((w, h, int(rng.integers(1, 12)), float(np.float32(rng.uniform(0.35, 0.95))), int(rng.integers(1, 4)),
                      int(rng.integers(1, 8)), int(rng.choice([3, 5, 7])), float(rng.choice([0.0, 0.45, 1.0, 1.5, 2.2])),
                      constancy, float(np.float32(10.0 ** rng.uniform(0.0, 2.0))), int(rng.integers(0, 1 << 30))))
    return cases


@pytest.mark.parametrize("w,h,levels,scale,outer,inner,median,sigma,constancy,alpha,seed", _random_cases(28, 20261003))
def test_randomized_pipelines_match_the_oracle(flow2d, oracle, make_flow, w, h, levels, scale, outer, inner, median, sigma,
                                               constancy, alpha, seed):
    """Whole ComputeFlow runs on configurations nobody hand-picked (AUTO therefore mixes the tiled, fused, chunked-fused,
    per-sweep and single-workgroup kernels across the levels of one run): bit-identical to the oracle.  Gradient cases
    are compared on whatever level sizes come out -- the oracle defines the reference's unwritten shared-memory slot
    the same way the product does."""
    f0, f1 = oracle.synthetic_pair(w, h, 1.0 + (seed % 7) * 0.3, -1.0 + (seed % 5) * 0.4, seed=seed, noise=True)
    flow = make_flow(w, h, constancy)
    u, v, _ = flow.compute_flow(f0, f1, flow.params(levels, scale, outer, inner, alpha, 0.001, 0.001, median, sigma))
    ou, ov, _ = oracle.compute_flow(f0, f1, levels, scale, outer, inner, alpha, 0.001, 0.001, median, sigma, constancy)
    assert np.array_equal(u, ou, equal_nan=True) and np.array_equal(v, ov, equal_nan=True)


def test_device_resident_entry_matches_host_entry(flow2d, oracle, make_flow, ctx):
    w, h = 200, 120
    f0, f1 = oracle.synthetic_pair(w, h, 2.0, 1.0, seed=3, noise=True)
    flow = flow2d.OpticalFlow(w, h, 0, ctx=ctx)
    try:
        p = flow.params(4, 0.5, 3, 5, 35.0, 0.001, 0.001, 5, 1.5)
        planes = [ctx.plane(w, h, a) for a in (f0, f1)] + [ctx.plane(w, h), ctx.plane(w, h)]
        assert planes[0].pitch == flow.pitch
        flow.compute_flow_device(*[pl.ptr for pl in planes], p, timing_mode=2)
        ctx.synchronize()
        u, v = planes[2].download(), planes[3].download()
        times = flow.level_timings()
        ou, ov, _ = oracle.compute_flow(f0, f1, 4, 0.5, 3, 5, 35.0, 0.001, 0.001, 5, 1.5)
        assert np.array_equal(u, ou) and np.array_equal(v, ov)
        assert [t[:2] for t in times] == [(25, 15), (50, 30), (100, 60), (200, 120)]
        # every level through the tiled kernel: one launch per outer iteration (3 here)
        assert all(t[2] > 0 for t in times) and [t[4] for t in times] == [3, 3, 3, 3]
        assert all(t[6] == flow2d.SOLVER_TILED for t in times)  # the record names the algorithm that ran
        assert all(t[3] < 0 for t in times[:-1]) and 0 < times[-1][3] <= times[-1][2] * 1.05  # launches timed on the finest level only
        assert times[-1][5] == (32.0 + 40.0 * 5) * 200 * 120  # algorithmic bytes of a fused launch (inner 5)
        # input frames are left untouched
        assert np.array_equal(planes[0].download(), f0) and np.array_equal(planes[1].download(), f1)
    finally:
        flow.close()


def test_full_size_properties(flow2d, oracle, make_flow):
    """Config 2 at full size (1024^2, 5 levels, 10 x 5 sweeps): too slow for a bit-compare against the
    single-thread path in a unit test budget?  No -- the OpenMP oracle does it in seconds, so compare in
    full, and also check size-independent properties: determinism and mirror symmetry of the solver."""
    w = h = 1024
    f0, f1 = oracle.synthetic_pair(w, h, 1.5, -0.75, seed=1)
    flow = make_flow(w, h)
    p = flow.params(5, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5)
    u, v, _ = flow.compute_flow(f0, f1, p)
    u2, v2, _ = flow.compute_flow(f0, f1, p)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)          # run-to-run bitwise reproducible
    ou, ov, _ = oracle.compute_flow(f0, f1, 5, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)
    assert np.isfinite(u).all() and np.isfinite(v).all()


def test_boundary_error_behaviour(flow2d, make_flow):
    flow = make_flow(64, 48)
    # a missing bag key: print + return, outputs untouched (optical_flow_2d.cpp:160-168)
    for key in ("warp_levels_count", "equation_alpha", "gaussian_sigma"):
        assert flow.missing_key_leaves_outputs(key) == 1
    # no usable level (scale >= 1) and a median width the operator refuses are input errors: ComputeFlow prints and
    # leaves the caller's flow untouched; the facade reports that no flow was delivered
    f = np.zeros((48, 64), np.float32)
    for bad in (flow.params(3, 1.0, 1, 1, 3.5, 0.001, 0.001, 5, 0.45), flow.params(3, 0.5, 1, 1, 3.5, 0.001, 0.001, 9, 0.45)):
        with pytest.raises(flow2d.Flow2DError) as e:
            flow.compute_flow(f, f, bad)
        assert e.value.status == 2


def test_operator_failure_fails_the_run(flow2d, oracle, make_flow, ctx):
    """A failing operator must fail the run, not hand back stale planes (the reference's Execute() is void and its
    ComputeFlow swaps the untouched output in).  Two reachable triggers: the single-workgroup solver on a level
    larger than 64 x 64, and a Gaussian longer than the 51 taps the kernel table holds (sigma >= 8.67)."""
    w, h = 200, 120
    f0, f1 = oracle.synthetic_pair(w, h, 2.0, 1.0, seed=3, noise=True)
    flow = make_flow(w, h)
    good = flow.params(3, 0.5, 2, 3, 35.0, 0.001, 0.001, 5, 1.5)
    for bad in (flow.params(3, 0.5, 2, 3, 35.0, 0.001, 0.001, 5, 1.5, flow2d.SOLVER_SINGLE_WORKGROUP),
                flow.params(3, 0.5, 2, 3, 35.0, 0.001, 0.001, 5, 9.0)):
        with pytest.raises(flow2d.Flow2DError):
            flow.compute_flow(f0, f1, bad)
    # the object is still usable afterwards and the pool is intact
    u, v, _ = flow.compute_flow(f0, f1, good)
    ou, ov, _ = oracle.compute_flow(f0, f1, 3, 0.5, 2, 3, 35.0, 0.001, 0.001, 5, 1.5)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)
    # device entry with graph replay: the broken run is reported on every call and never replayed from a cached graph
    dflow = flow2d.OpticalFlow(w, h, 0, ctx=ctx)
    try:
        planes = [ctx.plane(w, h, a) for a in (f0, f1)] + [ctx.plane(w, h).fill_bytes(0x7f), ctx.plane(w, h).fill_bytes(0x7f)]
        dflow.use_graph(True)
        bad = dflow.params(3, 0.5, 2, 3, 35.0, 0.001, 0.001, 5, 1.5, flow2d.SOLVER_SINGLE_WORKGROUP)
        for _ in range(2):
            with pytest.raises(flow2d.Flow2DError):
                dflow.compute_flow_device(*[p.ptr for p in planes], bad)
        ctx.synchronize()
        dflow.compute_flow_device(*[p.ptr for p in planes], good)
        dflow.compute_flow_device(*[p.ptr for p in planes], good)  # replayed
        ctx.synchronize()
        assert np.array_equal(planes[2].download(), ou) and np.array_equal(planes[3].download(), ov)
    finally:
        dflow.close()


def test_cli_rub_settings_file(flow2d, oracle, tmp_path):
    """The `flow2d` binary with the shipped rub settings file: exit code 0 and the reference's four
    output files (main.cpp:205-213), flow raws identical to the oracle."""
    out = tmp_path / "out"
    out.mkdir()
    xml = open(os.path.join(ROOT, "cuda-flow2d_amd", "host", "settings_rub.xml")).read()
    xml = xml.replace("./tests/data/", os.path.join(ROOT, "tests", "data") + "/").replace("./gpurun_out/", str(out) + "/")
    xml = xml.replace('levels="20"', 'levels="6"').replace('outer="20"', 'outer="2"')
    s = tmp_path / "settings.xml"
    s.write_text(xml)
    assert subprocess.call([flow2d.CLI_PATH, str(s)], stdout=subprocess.DEVNULL) == 0
    u = np.fromfile(out / "flow-u-584-388.raw", np.float32).reshape(388, 584)
    v = np.fromfile(out / "flow-v-584-388.raw", np.float32).reshape(388, 584)
    r1, r2 = rub_pair()
    ou, ov, _ = oracle.compute_flow(r1, r2, 6, 0.9, 2, 5, 3.5, 0.001, 0.001, 5, 0.45)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)
    amp = np.fromfile(out / "amp-584-388.raw", np.float32).reshape(388, 584)
    assert np.array_equal(amp, np.sqrt(u * u + v * v))
    ppm = open(out / "res.pgm", "rb").read()
    assert ppm.startswith(b"P6 \n584 388 \n255\n") and len(ppm) == 17 + 584 * 388 * 3
    # argv form: file1 file2 width height prefix outdir/  (+ --u8 superset); exit codes 2 and 3
    d = os.path.join(ROOT, "tests", "data")
    assert subprocess.call([flow2d.CLI_PATH, "--u8", d + "/rub1.raw", d + "/missing.raw", "584", "388", "x_",
                            str(out) + "/"], stdout=subprocess.DEVNULL) == 2
    assert subprocess.call([flow2d.CLI_PATH, str(tmp_path / "absent.xml")], stdout=subprocess.DEVNULL) == 3
    assert subprocess.call([flow2d.CLI_PATH, "a", "b", "c"], stdout=subprocess.DEVNULL) == 0  # usage


def test_config3_full_size_parity(flow2d, oracle, make_flow):
    """Config 3 as benchmarked (4096^2, Gradient, 8 levels, 10 x 5 sweeps, median 5, sigma 1.5): every pixel of the
    flow bit-identical to the oracle (OpenMP on the box's cores, some tens of seconds)."""
    w = h = 4096
    f0, f1 = oracle.synthetic_pair(w, h, 2.0, 1.0, seed=3)
    flow = make_flow(w, h, 1)
    u, v, _ = flow.compute_flow(f0, f1, flow.params(8, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5))
    ou, ov, _ = oracle.compute_flow(f0, f1, 8, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5, 1)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)


def test_config4_frame_size_parity(flow2d, oracle, make_flow):
    """One pair of config 4's shape (1920 x 1080, Grey, 8 levels: odd level sizes down to 15 x 9)."""
    w, h = 1920, 1080
    f0, f1 = oracle.synthetic_pair(w, h, 2.0 * np.cos(3.0), 2.0 * np.sin(3.0), seed=3)
    flow = make_flow(w, h)
    u, v, _ = flow.compute_flow(f0, f1, flow.params(8, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5))
    ou, ov, _ = oracle.compute_flow(f0, f1, 8, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)


def test_config4_batch_on_four_streams_with_graph_replay(flow2d, oracle):
    """Config 4 the way bench.py runs it: 8 distinct 1920 x 1080 pairs through OpticalFlowBatch2D (the C++ batch entry)
    on 4 lanes -- 2 pairs per lane, every lane with its own stream and OpticalFlow2D replaying recorded HIP graphs,
    all in flight together.  Every pair of the third, replayed round is bit-identical to the oracle."""
    w, h = 1920, 1080
    p = (8, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5)
    pairs = [oracle.synthetic_pair(w, h, 2.0 * np.cos(k), 2.0 * np.sin(k)) for k in range(8)]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, flow2d.GREY, lanes=4)
    try:
        assert batch.lanes == 4 and batch.pitch == c.plane(w, h).pitch
        planes = [(c.plane(w, h, f0), c.plane(w, h, f1), c.plane(w, h).fill_bytes(0x7f), c.plane(w, h).fill_bytes(0x7f))
                  for f0, f1 in pairs]
        columns = [[q[i].ptr for q in planes] for i in range(4)]
        params = batch.params(*p)
        for rnd in range(3):  # round 0 records the graphs, rounds 1 and 2 replay them
            if rnd == 2:
                batch.synchronize()
                for _, _, pu, pv in planes:
                    pu.fill_bytes(0x7f)
                    pv.fill_bytes(0x7f)
                c.synchronize()
            batch.compute_flow_batch_device(*columns, params)
        batch.synchronize()
        for k, (_, _, pu, pv) in enumerate(planes):
            ou, ov, _ = oracle.compute_flow(pairs[k][0], pairs[k][1], *p)
            assert np.array_equal(pu.download(), ou) and np.array_equal(pv.download(), ov), "pair %d" % k
    finally:
        batch.close()
        c.close()


@pytest.mark.parametrize("constancy", [0, 1])
def test_batch_entry_uneven_pairs_and_lane_offsets(flow2d, oracle, constancy):
    """OpticalFlowBatch2D: 5 pairs on 3 lanes (uneven), eager and graph-replayed, then the same pairs one call at a
    time with a rotating first lane: every flow equals the oracle, whichever lane and mode produced it."""
    w, h = 208, 144
    p = (4, 0.5, 3, 5, 35.0, 0.001, 0.001, 5, 1.5)
    pairs = [oracle.synthetic_pair(w, h, 1.5 * np.cos(k), -1.0 + 0.5 * k, seed=k, noise=True) for k in range(5)]
    want = [oracle.compute_flow(f0, f1, *p, constancy)[:2] for f0, f1 in pairs]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, constancy, lanes=3)
    try:
        planes = [(c.plane(w, h, f0), c.plane(w, h, f1), c.plane(w, h), c.plane(w, h)) for f0, f1 in pairs]
        columns = [[q[i].ptr for q in planes] for i in range(4)]
        params = batch.params(*p)

        def check(tag):
            batch.synchronize()
            for k, (_, _, pu, pv) in enumerate(planes):
                assert np.array_equal(pu.download(), want[k][0]) and np.array_equal(pv.download(), want[k][1]), (tag, k)
                pu.fill_bytes(0x7f)
                pv.fill_bytes(0x7f)
            c.synchronize()

        batch.use_graph(False)
        batch.compute_flow_batch_device(*columns, params)
        check("eager")
        batch.use_graph(True)
        for rnd in range(2):  # record, replay
            batch.compute_flow_batch_device(*columns, params)
            check("graph round %d" % rnd)
        for k in range(5):  # one pair per call, lanes 2, 0, 1, 2, 0
            batch.compute_flow_batch_device(*[[col[k]] for col in columns], params, first_lane=2 + k)
        check("single-pair calls")
        with pytest.raises(ValueError):
            batch.compute_flow_batch_device(columns[0], columns[1][:-1], columns[2], columns[3], params)
    finally:
        batch.close()
        c.close()


@pytest.mark.parametrize("w,h,G", [(208, 144, 3), (101, 75, 5), (512, 384, 3)])  # (512 x 384 x 3: AUTO gives the two finest levels of the group to the strips, the rest to the tiles)
@pytest.mark.parametrize("constancy,sigma,median", [(0, 1.5, 5), (1, 1.5, 5), (0, 0.0, 3), (3, 1.5, 5)])
def test_lock_step_groups_match_single_pairs(flow2d, oracle, constancy, sigma, median, w, h, G):
    """OpticalFlow2D::group_size (flow2d_context_set_batch): groups of 3 pairs stored one below the other in tall
    containers, every kernel launched once per group (grid.z); two groups on two lanes, eager and graph-replayed.
    Every pair of every group is bit-identical to the oracle's flow of that pair alone -- Grey, Gradient,
    LogDerivatives (per-instance launches of the single-workgroup kernel), with and without the pre-blur."""
    p = (4, 0.5, 3, 5, 35.0, 0.001, 0.001, median, sigma)
    pairs = [oracle.synthetic_pair(w, h, 1.5 * np.cos(k), -1.0 + 0.5 * k, seed=k, noise=True) for k in range(2 * G)]
    if constancy == 3:  # log(I + 1): the CPU libm and the device library differ in the last place; compare with the
        want = None     # product's own single-pair runs instead (those are pinned to the reference's kernel elsewhere)
    else:
        want = [oracle.compute_flow(f0, f1, *p, constancy)[:2] for f0, f1 in pairs]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, constancy, lanes=2, group_size=G)
    single = flow2d.OpticalFlow(w, h, constancy, ctx=c)
    try:
        assert batch.group_stride == batch.pitch * h
        if want is None:
            want = []
            for f0, f1 in pairs:
                pl = [c.plane(w, h, f0), c.plane(w, h, f1), c.plane(w, h), c.plane(w, h)]
                single.compute_flow_device(*[q.ptr for q in pl], single.params(*p))
                c.synchronize()
                want.append((pl[2].download(), pl[3].download()))
        groups = []
        for g in range(2):
            mine = pairs[g * G:(g + 1) * G]
            groups.append((c.plane(w, h * G, np.vstack([q[0] for q in mine])), c.plane(w, h * G, np.vstack([q[1] for q in mine])),
                           c.plane(w, h * G), c.plane(w, h * G)))
        columns = [[q[i].ptr for q in groups] for i in range(4)]
        params = batch.params(*p)
        for mode, rounds in (("eager", 1), ("graph", 2)):
            batch.use_graph(mode == "graph")
            for rnd in range(rounds):
                for _, _, pu, pv in groups:
                    pu.fill_bytes(0x7f)
                    pv.fill_bytes(0x7f)
                c.synchronize()
                batch.compute_flow_batch_device(*columns, params)
                batch.synchronize()
                for g, (_, _, pu, pv) in enumerate(groups):
                    u, v = pu.download(), pv.download()
                    for k in range(G):
                        wu, wv = want[g * G + k]
                        assert np.array_equal(u[k * h:(k + 1) * h], wu) and np.array_equal(v[k * h:(k + 1) * h], wv), \
                            (mode, rnd, g, k)
    finally:
        single.close()
        batch.close()
        c.close()


@pytest.mark.parametrize("constancy,inner", [(0, 7), (1, 5)])
def test_lock_step_group_of_chip_filling_pairs(flow2d, oracle, constancy, inner):
    """A group of two 4096 x 3456 pairs: on the finest level every instance fills the chip with strips of 128 rows and
    more, so the fused kernel is launched instance by instance (launch_fused_outer's split), while the coarser levels and
    the small kernels share launches.  Both pairs, eager and graph-replayed, equal the oracle's flow of each pair alone;
    a continued launch (7 sweeps = 4 + 3) goes through the same split."""
    w, h, G = 4096, 3456, 2
    for _ in range(1):
        p = (3, 0.5, 2, inner, 35.0, 0.001, 0.001, 5, 1.5)
        pairs = [oracle.synthetic_pair(w, h, 1.5 - k, 0.75 * k, seed=k + 1, noise=False) for k in range(G)]
        want = [oracle.compute_flow(f0, f1, *p, constancy)[:2] for f0, f1 in pairs]
        c = flow2d.Context(0)
        batch = flow2d.OpticalFlowBatch(w, h, constancy, lanes=1, group_size=G)
        try:
            group = (c.plane(w, h * G, np.vstack([q[0] for q in pairs])), c.plane(w, h * G, np.vstack([q[1] for q in pairs])),
                     c.plane(w, h * G), c.plane(w, h * G))
            columns = [[q.ptr] for q in group]
            params = batch.params(*p)
            for mode in ("eager", "graph"):
                batch.use_graph(mode == "graph")
                group[2].fill_bytes(0x7f)
                group[3].fill_bytes(0x7f)
                c.synchronize()
                batch.compute_flow_batch_device(*columns, params)
                batch.synchronize()
                u, v = group[2].download(), group[3].download()
                for k in range(G):
                    assert np.array_equal(u[k * h:(k + 1) * h], want[k][0]) and np.array_equal(v[k * h:(k + 1) * h], want[k][1]), \
                        (inner, mode, k)
        finally:
            batch.close()
            c.close()


@pytest.mark.parametrize("what", ["sor", "per-sweep", "seven sweeps"])
def test_lock_step_groups_on_the_unbatched_kernels(flow2d, oracle, what):
    """The solver paths that have no batched kernel run once per instance of a group (red-black SOR, the per-sweep
    kernels) or carry their continuation planes along (more than five sweeps: chunked fused launches on the larger
    level): each pair of a group of three still equals the oracle."""
    w, h, G = 672, 240, 3  # 672 x 240 and 336 x 120: the finest level goes to the strips when there are > 5 sweeps
    inner, kw, okw = 4, {}, {}
    if what == "sor":
        kw, okw = {"sor_omega": 1.4}, {"sor_omega": 1.4}
    elif what == "per-sweep":
        kw = {"algorithm": flow2d.SOLVER_PER_SWEEP}
    else:
        inner = 7
    p = (2, 0.5, 2, inner, 35.0, 0.001, 0.001, 5, 1.5)
    pairs = [oracle.synthetic_pair(w, h, 1.0 + 0.5 * k, -0.5 * k, seed=40 + k, noise=True) for k in range(G)]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, flow2d.GREY, lanes=1, group_size=G)
    try:
        planes = [c.plane(w, h * G, np.vstack([q[0] for q in pairs])), c.plane(w, h * G, np.vstack([q[1] for q in pairs])),
                  c.plane(w, h * G), c.plane(w, h * G)]
        batch.use_graph(False)
        batch.compute_flow_batch_device(*[[q.ptr] for q in planes], batch.params(*p, **kw))
        batch.synchronize()
        u, v = planes[2].download(), planes[3].download()
        for k, (f0, f1) in enumerate(pairs):
            ou, ov, _ = oracle.compute_flow(f0, f1, *p, flow2d.GREY, **okw)
            assert np.array_equal(u[k * h:(k + 1) * h], ou) and np.array_equal(v[k * h:(k + 1) * h], ov), k
    finally:
        batch.close()
        c.close()


def test_sequence_as_a_lock_step_group(flow2d, oracle):
    """An image sequence stored as one tall container needs no entry of its own in group mode: the group's frame-1 plane
    is its frame-0 plane one container further down, so the G flows of G + 1 consecutive frames are one
    ComputeFlowDevice call (frames are only read).  Each flow equals the oracle's flow of that pair."""
    w, h, G = 240, 136, 4
    p = (4, 0.5, 3, 5, 35.0, 0.001, 0.001, 5, 1.5)
    frames = [oracle.synthetic_pair(w, h, 0.8 * t, -0.4 * t, seed=9, noise=False)[1] for t in range(G + 1)]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, flow2d.GREY, lanes=1, group_size=G)
    try:
        tall = c.plane(w, h * (G + 1), np.vstack(frames))
        pu, pv = c.plane(w, h * G), c.plane(w, h * G)
        batch.compute_flow_batch_device([tall.ptr], [tall.ptr + batch.group_stride], [pu.ptr], [pv.ptr], batch.params(*p))
        batch.synchronize()
        u, v = pu.download(), pv.download()
        assert np.array_equal(tall.download(), np.vstack(frames))  # the sequence is left untouched
        for k in range(G):
            ou, ov, _ = oracle.compute_flow(frames[k], frames[k + 1], *p)
            assert np.array_equal(u[k * h:(k + 1) * h], ou) and np.array_equal(v[k * h:(k + 1) * h], ov), k
    finally:
        batch.close()
        c.close()


def test_config5_full_size_parity(flow2d, oracle, make_flow):
    """Config 5 as specified (8192^2, (12, -7) px shift, all 12 levels, 10 x 5 sweeps, median 5): every pixel of the flow
    bit-identical to the oracle (OpenMP on the box's cores: about half a minute and 3 GB of host memory)."""
    w = h = 8192
    f0, f1 = oracle.synthetic_pair(w, h, 12.0, -7.0, seed=5)
    flow = make_flow(w, h)
    u, v, _ = flow.compute_flow(f0, f1, flow.params(12, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5))
    ou, ov, _ = oracle.compute_flow(f0, f1, 12, 0.5, 10, 5, 35.0, 0.001, 0.001, 5, 1.5)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)
    assert np.isfinite(u).all() and np.isfinite(v).all()


def test_config5_size_properties(flow2d, oracle, make_flow):
    """Config 5's size (8192^2, all 12 levels) through size-independent properties as well: the fused path (AUTO) and
    the per-launch path (one launch per reference launch) agree bit for bit, runs are reproducible, outputs finite."""
    w = h = 8192
    f0, f1 = oracle.synthetic_pair(w, h, 12.0, -7.0, seed=5)
    flow = make_flow(w, h)
    u, v, _ = flow.compute_flow(f0, f1, flow.params(12, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5))
    u1, v1, _ = flow.compute_flow(f0, f1, flow.params(12, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5, flow2d.SOLVER_PER_SWEEP))
    assert np.array_equal(u, u1) and np.array_equal(v, v1)
    u2, v2, _ = flow.compute_flow(f0, f1, flow.params(12, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5))
    assert np.array_equal(u, u2) and np.array_equal(v, v2)
    assert np.isfinite(u).all() and np.isfinite(v).all()


@pytest.mark.parametrize("w,h,levels,outer,inner,sigma,constancy", [
    (200, 120, 4, 2, 3, 1.5, 0),     # pre-blur: every frame blurred and resampled once
    (200, 120, 4, 2, 3, 0.0, 1),     # no pre-blur: the callers' planes are level 0 themselves
    (640, 528, 3, 1, 5, 1.5, 1),     # a fused-kernel level
])
def test_sequence_matches_pairwise(flow2d, oracle, ctx, w, h, levels, outer, inner, sigma, constancy):
    """ComputeFlowSequenceDevice: flow k of the sequence is bit-identical to the pair (k, k+1) computed on its
    own (and the first one to the oracle); frames are left untouched; a second call reuses the cache planes."""
    flow = flow2d.OpticalFlow(w, h, constancy, ctx=ctx)
    try:
        p = flow.params(levels, 0.5, outer, inner, 35.0, 0.001, 0.001, 5, sigma)
        shifts = [(0.0, 0.0), (1.5, -0.75), (2.5, 0.5), (4.0, 1.0), (3.0, 2.5)]
        frames = [oracle.synthetic_pair(w, h, dx, dy, seed=7 + k, noise=True)[1] for k, (dx, dy) in enumerate(shifts)]
        planes = [ctx.plane(w, h, f) for f in frames]
        n = len(frames)
        us, vs = [ctx.plane(w, h) for _ in range(n - 1)], [ctx.plane(w, h) for _ in range(n - 1)]
        pu, pv = ctx.plane(w, h), ctx.plane(w, h)
        for rep in range(2):
            for pl in us + vs:
                pl.fill_bytes(0x33)
            flow.compute_flow_sequence_device([pl.ptr for pl in planes], [pl.ptr for pl in us], [pl.ptr for pl in vs], p)
            ctx.synchronize()
            for k in range(n - 1):
                flow.compute_flow_device(planes[k].ptr, planes[k + 1].ptr, pu.ptr, pv.ptr, p)
                ctx.synchronize()
                assert np.array_equal(us[k].download(), pu.download()), "u of pair %d" % k
                assert np.array_equal(vs[k].download(), pv.download()), "v of pair %d" % k
        ou, ov, _ = oracle.compute_flow(frames[0], frames[1], levels, 0.5, outer, inner, 35.0, 0.001, 0.001, 5, sigma,
                                        constancy)
        assert np.array_equal(us[0].download(), ou) and np.array_equal(vs[0].download(), ov)
        for pl, f in zip(planes, frames):
            assert np.array_equal(pl.download(), f)
    finally:
        flow.close()


def test_graph_replay_matches_eager(flow2d, oracle, ctx):
    """A recorded pyramid replayed on new frame contents gives the same bits as eager launches / the oracle."""
    w, h = 160, 96
    flow = flow2d.OpticalFlow(w, h, 0, ctx=ctx)
    try:
        p = flow.params(4, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5)
        f0, f1 = oracle.synthetic_pair(w, h, 1.0, 0.5, seed=2, noise=True)
        planes = [ctx.plane(w, h, f0), ctx.plane(w, h, f1), ctx.plane(w, h), ctx.plane(w, h)]
        ptrs = [pl.ptr for pl in planes]
        flow.compute_flow_device(*ptrs, p)
        ctx.synchronize()
        eager_u, eager_v = planes[2].download(), planes[3].download()
        flow.use_graph(True)
        for _ in range(3):  # first call records, the others replay
            planes[2].fill_bytes(0x55)
            planes[3].fill_bytes(0x55)
            flow.compute_flow_device(*ptrs, p)
            ctx.synchronize()
            assert np.array_equal(planes[2].download(), eager_u) and np.array_equal(planes[3].download(), eager_v)
        # same buffers, new contents: the replayed graph must compute the new pair
        g0, g1 = oracle.synthetic_pair(w, h, -1.5, 0.75, seed=4, noise=True)
        planes[0].upload(g0)
        planes[1].upload(g1)
        flow.compute_flow_device(*ptrs, p)
        ctx.synchronize()
        ou, ov, _ = oracle.compute_flow(g0, g1, 4, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5)
        assert np.array_equal(planes[2].download(), ou) and np.array_equal(planes[3].download(), ov)
        # different parameters -> a different graph, not a stale replay
        p2 = flow.params(3, 0.5, 1, 3, 35.0, 0.001, 0.001, 3, 0.45)
        flow.compute_flow_device(*ptrs, p2)
        ctx.synchronize()
        ou, ov, _ = oracle.compute_flow(g0, g1, 3, 0.5, 1, 3, 35.0, 0.001, 0.001, 3, 0.45)
        assert np.array_equal(planes[2].download(), ou) and np.array_equal(planes[3].download(), ov)
    finally:
        flow.close()


@pytest.mark.parametrize("constancy", [0, 1])
@pytest.mark.parametrize("w,h,levels", [(512, 384, 6), (200, 136, 4), (1024, 1024, 7), (250, 150, 5)])
def test_levels_resampled_up_front_match_the_oracle(flow2d, oracle, ctx, w, h, levels, constancy):
    """Round 6: with a scale factor of 0.5 every level of both frames is resampled up front -- the x passes of all levels in one
    launch, the y passes of all levels in one launch -- into plane regions one below the other, and a level's warp writes a plane of
    its own instead of replacing the level's frame.  Same cell sums: the flow equals the oracle's bit for bit, eager and replayed, for
    a lone object (packed strip build on under-filled launches) and a pipeline lane's, and when pairs are queued back to back."""
    p = flow2d.OpticalFlow.params(levels, 0.5, 3, 5, 35.0, 0.001, 0.001, 5, 1.5)
    pairs = [oracle.synthetic_pair(w, h, 1.25 + k, -0.5 * k, seed=20 + k, noise=True) for k in range(3)]
    wanted = [oracle.compute_flow(f0, f1, levels, 0.5, 3, 5, 35.0, 0.001, 0.001, 5, 1.5, constancy)[:2] for f0, f1 in pairs]
    for lone in (True, False):
        flow = flow2d.OpticalFlow(w, h, constancy, ctx=ctx, lone=lone)
        try:
            planes = [ctx.plane(w, h), ctx.plane(w, h), ctx.plane(w, h), ctx.plane(w, h)]
            outs = [(ctx.plane(w, h), ctx.plane(w, h)) for _ in pairs]
            for graph in (False, True):
                flow.use_graph(graph)
                for (f0, f1), (ou, ov) in zip(pairs, wanted):
                    planes[0].upload(f0)
                    planes[1].upload(f1)
                    planes[2].fill_bytes(0x55)
                    planes[3].fill_bytes(0x55)
                    flow.compute_flow_device(*[pl.ptr for pl in planes], p)
                    ctx.synchronize()
                    assert np.array_equal(planes[2].download(), ou) and np.array_equal(planes[3].download(), ov), (lone, graph)
                # three pairs queued back to back without a host wait, each into flow planes of its own
                frames = [(ctx.plane(w, h, f0), ctx.plane(w, h, f1)) for f0, f1 in pairs]
                for (a, b), (u, v) in zip(frames, outs):
                    u.fill_bytes(0x55)
                    v.fill_bytes(0x55)
                    flow.compute_flow_device(a.ptr, b.ptr, u.ptr, v.ptr, p)
                ctx.synchronize()
                for (u, v), (ou, ov) in zip(outs, wanted):
                    assert np.array_equal(u.download(), ou) and np.array_equal(v.download(), ov), (lone, graph, "back to back")
        finally:
            flow.close()
    ctx.set_lone(False)


def test_graph_cache_evicts_the_least_recently_used(flow2d, oracle, ctx):
    """More (buffers, parameters) combinations than the cache of recorded pyramids holds (32): the least recently replayed
    graph goes, one at a time, and whatever is replayed or re-recorded afterwards still computes the right flow."""
    w, h = 64, 48
    flow = flow2d.OpticalFlow(w, h, 0, ctx=ctx)
    try:
        f0, f1 = oracle.synthetic_pair(w, h, 1.0, 0.5, seed=2, noise=True)
        planes = [ctx.plane(w, h, f0), ctx.plane(w, h, f1), ctx.plane(w, h), ctx.plane(w, h)]
        ptrs = [pl.ptr for pl in planes]
        flow.use_graph(True)
        want = {}
        for rnd in range(2):
            for k in range(36):  # 36 parameter sets through a cache of 32, twice
                alpha = 5.0 + k
                if k not in want:
                    want[k] = oracle.compute_flow(f0, f1, 3, 0.5, 1, 2, alpha, 0.001, 0.001, 3, 0.45)[:2]
                planes[2].fill_bytes(0x55)
                planes[3].fill_bytes(0x55)
                flow.compute_flow_device(*ptrs, flow.params(3, 0.5, 1, 2, alpha, 0.001, 0.001, 3, 0.45))
                ctx.synchronize()
                assert np.array_equal(planes[2].download(), want[k][0]) and np.array_equal(planes[3].download(), want[k][1]), (rnd, k)
    finally:
        flow.close()


def test_graph_replay_with_chunked_fused_solver(flow2d, oracle, ctx):
    """7 sweeps per outer iteration at a fused-kernel level: two launches per outer iteration and, with one outer
    iteration, the hand-over copy out of the third plane pair -- all inside a recorded graph."""
    w, h = 640, 528
    flow = flow2d.OpticalFlow(w, h, 0, ctx=ctx)
    try:
        p = flow.params(2, 0.5, 1, 7, 35.0, 0.001, 0.001, 5, 1.5)
        f0, f1 = oracle.synthetic_pair(w, h, 1.0, 0.5, seed=6, noise=True)
        planes = [ctx.plane(w, h, f0), ctx.plane(w, h, f1), ctx.plane(w, h), ctx.plane(w, h)]
        ptrs = [pl.ptr for pl in planes]
        ou, ov, _ = oracle.compute_flow(f0, f1, 2, 0.5, 1, 7, 35.0, 0.001, 0.001, 5, 1.5)
        flow.use_graph(True)
        for _ in range(3):
            planes[2].fill_bytes(0x55)
            planes[3].fill_bytes(0x55)
            flow.compute_flow_device(*ptrs, p)
            ctx.synchronize()
            assert np.array_equal(planes[2].download(), ou) and np.array_equal(planes[3].download(), ov)
    finally:
        flow.close()


@pytest.mark.parametrize("constancy", [0, 1, 2])
def test_opt_in_sor_pyramid(flow2d, oracle, make_flow, constancy):
    """The opt-in red-black SOR mode end to end (bag key solver_sor_omega) against its oracle restatement.
    This mode has no counterpart in the reference (Jacobi), so it is not part of the reference-parity claim."""
    w, h = 192, 128
    f0, f1 = oracle.synthetic_pair(w, h, 1.5, -0.75, seed=6, noise=True)
    flow = make_flow(w, h, constancy)
    p = flow.params(4, 0.5, 3, 4, 35.0, 0.001, 0.001, 5, 1.5, sor_omega=1.4)
    u, v, _ = flow.compute_flow(f0, f1, p)
    ou, ov, _ = oracle.compute_flow(f0, f1, 4, 0.5, 3, 4, 35.0, 0.001, 0.001, 5, 1.5, constancy, sor_omega=1.4)
    assert np.array_equal(u, ou) and np.array_equal(v, ov)
    ju, jv, _ = oracle.compute_flow(f0, f1, 4, 0.5, 3, 4, 35.0, 0.001, 0.001, 5, 1.5, constancy)
    assert not np.array_equal(ou, ju)  # it really is a different relaxation


@pytest.mark.parametrize("constancy,sigma", [(0, 1.5), (1, 1.5), (0, 0.0), (3, 1.5)])
def test_groups_formed_from_scattered_planes(flow2d, oracle, constancy, sigma):
    """OpticalFlowBatch2D::ComputeFlowBatchDeviceGrouped: eleven independent pairs, every plane an allocation of its own,
    through two lanes in groups of four (4 + 4 + 3: the last group is smaller).  The object gathers the frames into its
    tall staging containers, computes each group with one launch per kernel and hands the flows back; every pair equals
    the oracle, eager and replayed from graphs, and the frames are left untouched."""
    w, h, G, n = (208, 144, 4, 11) if constancy != 3 else (208, 144, 4, 5)
    alpha = 35.0 if constancy != 3 else 0.0005
    p = (3, 0.5, 2, 5, alpha, 0.001, 0.001, 5, sigma)
    pairs = [oracle.synthetic_pair(w, h, 1.0 + 0.25 * k, -0.5 + 0.2 * k, seed=120 + k, noise=True) for k in range(n)]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, constancy, lanes=2, group_size=G)
    try:
        f0s = [c.plane(w, h, q[0]) for q in pairs]
        f1s = [c.plane(w, h, q[1]) for q in pairs]
        us = [c.plane(w, h) for _ in pairs]
        vs = [c.plane(w, h) for _ in pairs]
        c.synchronize()
        if constancy == 3:  # the CPU's logf differs from the device's in the last place: Log pairs against single-pair runs
            single = flow2d.OpticalFlow(w, h, constancy, ctx=c)
            want = []
            for k in range(n):
                single.compute_flow_device(f0s[k].ptr, f1s[k].ptr, us[k].ptr, vs[k].ptr, single.params(*p))
                c.synchronize()
                want.append((us[k].download(), vs[k].download()))
            single.close()
        else:
            want = [oracle.compute_flow(f0, f1, *p, constancy)[:2] for f0, f1 in pairs]
        for graph in (False, True, True):
            batch.use_graph(graph)
            for q in us + vs:
                q.fill_bytes(0x7f)
            c.synchronize()
            batch.compute_flow_batch_device_grouped([q.ptr for q in f0s], [q.ptr for q in f1s], [q.ptr for q in us],
                                                    [q.ptr for q in vs], batch.params(*p))
            batch.synchronize()
            for k in range(n):
                assert np.array_equal(us[k].download(), want[k][0]) and np.array_equal(vs[k].download(), want[k][1]), (graph, k)
                assert np.array_equal(f0s[k].download(), pairs[k][0]) and np.array_equal(f1s[k].download(), pairs[k][1])
    finally:
        batch.close()
        c.close()


@pytest.mark.parametrize("constancy", [0, 1])
def test_groups_run_in_place_when_the_pairs_sit_one_container_apart(flow2d, oracle, constancy):
    """ComputeFlowBatchDeviceGrouped with eleven pairs whose planes lie one below the other in four allocations per group of four
    (4 + 4 + 3), handed over pair by pair like independent planes: a run of pairs exactly GroupStrideBytes() apart in all four roles
    is a group as it lies -- the pyramid runs on the caller's planes, nothing is gathered or handed back (round 6).  Every pair
    equals the oracle, eager and replayed, frames untouched; a run with ONE role out of step (the u planes of a group swapped) still
    goes through the staging containers, to the same flows."""
    w, h, G, n = 208, 144, 4, 11
    p = (3, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5)
    pairs = [oracle.synthetic_pair(w, h, 1.0 + 0.25 * k, -0.5 + 0.2 * k, seed=320 + k, noise=True) for k in range(n)]
    want = [oracle.compute_flow(f0, f1, *p, constancy)[:2] for f0, f1 in pairs]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, constancy, lanes=2, group_size=G)
    try:
        stride = batch.group_stride
        groups = [pairs[i:i + G] for i in range(0, n, G)]
        tall = []
        for g in groups:  # (the last group's allocation is four pairs tall too: its fourth slot stays unused)
            pad = g + [g[-1]] * (G - len(g))
            tall.append([c.plane(w, h * G, np.vstack([q[0] for q in pad])), c.plane(w, h * G, np.vstack([q[1] for q in pad])),
                         c.plane(w, h * G), c.plane(w, h * G)])
        c.synchronize()
        assert tall[0][0].pitch * h == stride

        def pointers(role, swap_first_two_of_group=None):
            out = []
            for gi, g in enumerate(groups):
                ptrs = [tall[gi][role].ptr + k * stride for k in range(len(g))]
                if swap_first_two_of_group == gi:
                    ptrs[0], ptrs[1] = ptrs[1], ptrs[0]
                out += ptrs
            return out

        def flows():
            us, vs = [], []
            for gi, g in enumerate(groups):
                u, v = tall[gi][2].download(), tall[gi][3].download()
                us += [u[k * h:(k + 1) * h] for k in range(len(g))]
                vs += [v[k * h:(k + 1) * h] for k in range(len(g))]
            return us, vs

        for graph in (False, True, True):
            batch.use_graph(graph)
            for t in tall:
                t[2].fill_bytes(0x7f), t[3].fill_bytes(0x7f)
            c.synchronize()
            batch.compute_flow_batch_device_grouped(pointers(0), pointers(1), pointers(2), pointers(3), batch.params(*p))
            batch.synchronize()
            us, vs = flows()
            for k in range(n):
                assert np.array_equal(us[k], want[k][0]) and np.array_equal(vs[k], want[k][1]), (graph, k)
            # the unused fourth slot of the last group was not written
            assert np.all(tall[-1][2].download()[len(groups[-1]) * h:].view(np.uint32) == 0x7f7f7f7f)
            for gi, g in enumerate(groups):
                f0, f1 = tall[gi][0].download(), tall[gi][1].download()
                for k, q in enumerate(g):
                    assert np.array_equal(f0[k * h:(k + 1) * h], q[0]) and np.array_equal(f1[k * h:(k + 1) * h], q[1])
        # group 1's u planes out of step: pair 4 writes where pair 5's u lies and the other way round
        batch.use_graph(True)
        for t in tall:
            t[2].fill_bytes(0x7f), t[3].fill_bytes(0x7f)
        c.synchronize()
        batch.compute_flow_batch_device_grouped(pointers(0), pointers(1), pointers(2, swap_first_two_of_group=1), pointers(3),
                                                batch.params(*p))
        batch.synchronize()
        us, vs = flows()
        us[4], us[5] = us[5], us[4]
        for k in range(n):
            assert np.array_equal(us[k], want[k][0]) and np.array_equal(vs[k], want[k][1]), ("out of step", k)
    finally:
        batch.close()
        c.close()


def test_scattered_group_larger_than_one_gather_launch(flow2d, oracle):
    """Groups of more than 32 scattered pairs need more planes than one flow2d_copy_planes launch names (64): the gather
    and the hand-back are issued in chunks.  40 small pairs as ONE lock-step group, every pair against the oracle.
    (Round 3 accepted group sizes up to 64 at Initialize and then refused every such group.)"""
    w, h, G = 64, 48, 40
    p = (2, 0.5, 2, 3, 35.0, 0.001, 0.001, 3, 1.0)
    pairs = [oracle.synthetic_pair(w, h, 0.5 + 0.05 * k, -0.3 + 0.02 * k, seed=900 + k, noise=True) for k in range(G)]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, flow2d.GREY, lanes=1, group_size=G)
    try:
        f0s = [c.plane(w, h, q[0]) for q in pairs]
        f1s = [c.plane(w, h, q[1]) for q in pairs]
        us = [c.plane(w, h).fill_bytes(0x7f) for _ in pairs]
        vs = [c.plane(w, h).fill_bytes(0x7f) for _ in pairs]
        c.synchronize()
        for graph in (False, True, True):
            batch.use_graph(graph)
            batch.compute_flow_batch_device_grouped([q.ptr for q in f0s], [q.ptr for q in f1s], [q.ptr for q in us],
                                                    [q.ptr for q in vs], batch.params(*p))
            batch.synchronize()
            for k, (f0, f1) in enumerate(pairs):
                ou, ov, _ = oracle.compute_flow(f0, f1, *p, flow2d.GREY)
                assert np.array_equal(us[k].download(), ou) and np.array_equal(vs[k].download(), ov), (graph, k)
            for q in us + vs:
                q.fill_bytes(0x7f)
            c.synchronize()
    finally:
        batch.close()
        c.close()


def test_group_of_one_scattered_pair_and_the_tall_entry_do_not_share_a_graph(flow2d, oracle):
    """On an object with groups of two, ComputeFlowGroupDevice(count = 1) and ComputeFlowDevice may be handed the same
    four pointers: the first records gather -> pyramid of ONE instance -> hand back, the second a pyramid over the TWO
    instances of tall containers at those pointers.  Their graphs are keyed apart (entry tag + instance count); round 3
    replayed whichever had been recorded first for both."""
    w, h, G = 96, 64, 2
    p = (3, 0.5, 2, 4, 35.0, 0.001, 0.001, 5, 1.2)
    pairs = [oracle.synthetic_pair(w, h, 1.0 + 0.5 * k, -0.4 * k, seed=940 + k, noise=True) for k in range(G)]
    want = [oracle.compute_flow(f0, f1, *p, flow2d.GREY)[:2] for f0, f1 in pairs]
    c = flow2d.Context(0)
    batch = flow2d.OpticalFlowBatch(w, h, flow2d.GREY, lanes=1, group_size=G)
    try:
        # tall containers: pair g of every plane `group_stride` bytes behind the pointer
        tall = [c.plane(w, h * G, np.vstack([q[0] for q in pairs])), c.plane(w, h * G, np.vstack([q[1] for q in pairs])),
                c.plane(w, h * G), c.plane(w, h * G)]
        c.synchronize()
        batch.use_graph(True)
        for order in ("scattered first", "again"):
            # a scattered group of one pair: only the first instance's flow is written
            for q in tall[2:]:
                q.fill_bytes(0x7f)
            c.synchronize()
            batch.compute_flow_batch_device_grouped([tall[0].ptr], [tall[1].ptr], [tall[2].ptr], [tall[3].ptr], batch.params(*p))
            batch.synchronize()
            u, v = tall[2].download(), tall[3].download()
            assert np.array_equal(u[:h], want[0][0]) and np.array_equal(v[:h], want[0][1]), order
            assert np.all(u[h:].view(np.uint32) == 0x7f7f7f7f) and np.all(v[h:].view(np.uint32) == 0x7f7f7f7f), order
            # the tall entry on the same pointers: both instances
            batch.compute_flow_batch_device([tall[0].ptr], [tall[1].ptr], [tall[2].ptr], [tall[3].ptr], batch.params(*p))
            batch.synchronize()
            u, v = tall[2].download(), tall[3].download()
            for g in range(G):
                assert np.array_equal(u[g * h:(g + 1) * h], want[g][0]) and np.array_equal(v[g * h:(g + 1) * h], want[g][1]), (order, g)
    finally:
        batch.close()
        c.close()


def test_cli_full_settings_xml_values(flow2d, tmp_path):
    """The `flow2d` binary with the values of the reference's settings.xml untouched (20 levels at 0.9, 20 x 5 sweeps,
    median 5, sigma 0.45, alpha 3.5) on rub1/rub2: the flow raws it writes hash to what the reference's own kernels
    produced for this configuration on the MI355X (tests/golden/ref_kernels_golden.npz, `rub_settings`)."""
    import hashlib
    out = tmp_path / "out"
    out.mkdir()
    xml = open(os.path.join(ROOT, "cuda-flow2d_amd", "host", "settings_rub.xml")).read()
    xml = xml.replace("./tests/data/", os.path.join(ROOT, "tests", "data") + "/").replace("./gpurun_out/", str(out) + "/")
    assert 'levels="20"' in xml and 'outer="20"' in xml
    s = tmp_path / "settings.xml"
    s.write_text(xml)
    assert subprocess.call([flow2d.CLI_PATH, str(s)], stdout=subprocess.DEVNULL) == 0
    golden = np.load(os.path.join(ROOT, "tests", "golden", "ref_kernels_golden.npz"))
    want_u, want_v = (str(x) for x in golden["rub_settings_sha"])
    sha = lambda name: hashlib.sha256(np.fromfile(out / name, "<f4").tobytes()).hexdigest()
    assert sha("flow-u-584-388.raw") == want_u and sha("flow-v-584-388.raw") == want_v


def test_cli_unwritable_output_is_exit_code_255(flow2d, tmp_path):
    """The reference ends with exit(-1) -- status 255 -- when the colour-wheel image or the magnitude raw cannot be
    written (src/utils/io_utils.cpp:47-51,93-97): an output directory that does not exist does that, after the flow has
    been computed."""
    d = os.path.join(ROOT, "tests", "data")
    rc = subprocess.call([flow2d.CLI_PATH, "--u8", d + "/rub1.raw", d + "/rub2.raw", "584", "388", "x_",
                          str(tmp_path / "no" / "such" / "dir") + "/", "3.5", "0.45"], stdout=subprocess.DEVNULL,
                         stderr=subprocess.DEVNULL)
    assert rc == 255
