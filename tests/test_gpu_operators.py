"""The reference's operator ("plugin") API -- CudaOperation*2D::Initialize / Execute with a string-keyed bag of
void* (src/cuda_operations/**) -- exercised directly, with the reference's key names and pointee types, and
checked against the oracle.  Also the error behaviour: a missing key or in == out prints and leaves the
output untouched."""
import ctypes as C

import numpy as np
import pytest

from conftest import in_container, level_fields

pytestmark = pytest.mark.gpu

W, H, CW, CH = 100, 70, 128, 80


def dp(plane):
    return C.c_ulonglong(plane.ptr)


@pytest.fixture()
def rig(flow2d, ctx, oracle):
    planes = {}

    def up(name, a=None, fill=0.0):
        planes[name] = ctx.plane(CW, CH, in_container(a, CW, CH, fill) if a is not None else None)
        if a is None:
            planes[name].fill_bytes(0x7f)
        return planes[name]

    ops = []

    def make(kind, constancy=0, **kw):
        op = flow2d.Operator(kind, CW, CH, planes[next(iter(planes))].pitch if planes else ctx.plane(CW, CH).pitch,
                             constancy, ctx=ctx, **kw)
        ops.append(op)
        return op

    yield up, make
    for op in ops:
        op.close()
    flow2d.host_lib().flow2d_host_adopt_context(None)


def test_bag_semantics(flow2d):
    assert flow2d.host_lib().flow2d_host_bag_selftest() == 0


def test_operator_names_and_init(flow2d, rig):
    up, make = rig
    up("a")
    names = {k: make(k).name for k in ("add", "convolution", "median", "registration", "resample", "solve")}
    assert names == {"add": "CUDA Add 2D", "convolution": "CUDA Convolution 2D", "median": "CUDA Median 2D",
                     "registration": "CUDA Registration 2D", "resample": "CUDA Resample 2D", "solve": "CUDA Solve 2D"}
    with pytest.raises(flow2d.Flow2DError):  # container_size is mandatory at Initialize
        make("add", omit_container_size=True)
    assert make("solve", constancy=2).name == "CUDA Solve 2D"  # DataConstancy::LogDerivatives selects solve_2d_log


def test_add_convolution_median(flow2d, oracle, rig):
    up, make = rig
    f0, f1, u, *_ = level_fields(oracle, W, H, 21)
    size = flow2d.DataSize3(W, H, 0)
    a, b = up("a", f0), up("b", f1)
    make("add").execute(operand_0=dp(a), operand_1=dp(b), data_size=size)
    assert np.array_equal(a.download(W, H), oracle.add(f0, f1, W, H))

    src, dst, tmp = up("src", f0), up("dst"), up("tmp")
    make("convolution").execute(dev_input=dp(src), dev_output=dp(dst), dev_temp=dp(tmp), data_size=size,
                                gaussian_sigma=C.c_float(1.5))
    assert np.array_equal(dst.download(W, H), oracle.convolution(f0, W, H, 1.5))

    msrc, mdst = up("msrc", u), up("mdst")
    med = make("median")
    for radius, want in ((5, 5), (6, 5), (3, 3), (7, 7)):  # an even width is reduced by one with a warning
        med.execute(dev_input=dp(msrc), dev_output=dp(mdst), data_size=size, radius=C.c_size_t(radius))
        assert np.array_equal(mdst.download(W, H), oracle.median(u, W, H, want))
    med.execute(dev_input=dp(msrc), dev_output=dp(mdst), data_size=size, radius=C.c_size_t(1))  # width 1 = copy
    assert np.array_equal(mdst.download(), msrc.download())
    # unsupported width: error message, output untouched (the reference then swaps in the stale buffer)
    before = mdst.download()
    med.execute(dev_input=dp(msrc), dev_output=dp(mdst), data_size=size, radius=C.c_size_t(9))
    assert np.array_equal(mdst.download(), before)
    # in == out and a missing key: print and return
    med.execute(dev_input=dp(msrc), dev_output=dp(msrc), data_size=size, radius=C.c_size_t(5))
    med.execute(dev_input=dp(msrc), dev_output=dp(mdst), data_size=size)
    assert np.array_equal(mdst.download(), before)


def test_registration_and_resample(flow2d, oracle, rig):
    up, make = rig
    f0, f1, u, v, *_ = level_fields(oracle, W, H, 22, flow_scale=3.0)
    size = flow2d.DataSize3(W, H, 0)
    planes = [up(n, a) for n, a in (("f0", f0), ("f1", f1), ("u", u), ("v", v))]
    out = up("out")
    make("registration").execute(dev_frame_0=dp(planes[0]), dev_frame_1=dp(planes[1]), dev_flow_u=dp(planes[2]),
                                 dev_flow_v=dp(planes[3]), dev_output=dp(out), data_size=size, hx=C.c_float(1.25),
                                 hy=C.c_float(1.1))
    assert np.array_equal(out.download(W, H), oracle.registration(f0, f1, u, v, W, H, 1.25, 1.1))

    dst, tmp = up("dst"), up("tmp")
    new = flow2d.DataSize3(37, 20, 0)
    make("resample").execute(dev_input=dp(planes[0]), dev_output=dp(dst), dev_temp=dp(tmp), data_size=size,
                             resample_size=new)
    want = oracle.resample(in_container(f0, CW, CH), W, H, 37, 20)[:20, :37]
    assert np.array_equal(dst.download(37, 20), want)


def test_median_with_addend_and_resample_upwards(flow2d, oracle, rig):
    """The two one-launch forms the pyramid uses, through the operator bags: the median operator's optional dev_addend /
    dev_addend_b keys (filter of input + addend; the input planes stay as they were; width 1 = add, then copy) and the
    resample operator when both directions up-sample (both passes in one launch, dev_temp unused)."""
    up, make = rig
    _, _, u, v, *_ = level_fields(oracle, W, H, 23)
    rng = np.random.default_rng(3)
    du, dv = (rng.normal(0, 0.2, (H, W)).astype(np.float32) for _ in range(2))
    size = flow2d.DataSize3(W, H, 0)
    pu, pv, pdu, pdv, ou, ov = up("u", u), up("v", v), up("du", du), up("dv", dv), up("ou"), up("ov")
    med = make("median")
    for radius, want in ((5, 5), (4, 3), (7, 7)):
        med.execute(dev_input=dp(pu), dev_output=dp(ou), dev_input_b=dp(pv), dev_output_b=dp(ov), dev_addend=dp(pdu),
                    dev_addend_b=dp(pdv), data_size=size, radius=C.c_size_t(radius))
        assert np.array_equal(ou.download(W, H), oracle.median(oracle.add(u, du, W, H), W, H, want))
        assert np.array_equal(ov.download(W, H), oracle.median(oracle.add(v, dv, W, H), W, H, want))
        assert np.array_equal(pu.download(W, H), u) and np.array_equal(pv.download(W, H), v)
    single = up("single")
    med.execute(dev_input=dp(pu), dev_output=dp(single), dev_addend=dp(pdu), data_size=size, radius=C.c_size_t(3))
    assert np.array_equal(single.download(W, H), oracle.median(oracle.add(u, du, W, H), W, H, 3))
    # width 1: no filter to carry the sum, so the operator adds in place (like the reference's add) and copies
    med.execute(dev_input=dp(pu), dev_output=dp(ou), dev_input_b=dp(pv), dev_output_b=dp(ov), dev_addend=dp(pdu),
                dev_addend_b=dp(pdv), data_size=size, radius=C.c_size_t(1))
    assert np.array_equal(ou.download(W, H), oracle.add(u, du, W, H)) and np.array_equal(ov.download(W, H), oracle.add(v, dv, W, H))
    assert np.array_equal(pu.download(W, H), oracle.add(u, du, W, H))

    small = flow2d.DataSize3(37, 20, 0)
    a, b = u[:20, :37].copy(), v[:20, :37].copy()
    sa, sb, da, db, ta, tb = up("sa", a), up("sb", b), up("da"), up("db"), up("ta"), up("tb")
    make("resample").execute(dev_input=dp(sa), dev_output=dp(da), dev_temp=dp(ta), dev_input_b=dp(sb), dev_output_b=dp(db),
                             dev_temp_b=dp(tb), data_size=small, resample_size=size)
    assert np.array_equal(da.download(W, H), oracle.resample(in_container(a, CW, CH), 37, 20, W, H)[:H, :W])
    assert np.array_equal(db.download(W, H), oracle.resample(in_container(b, CW, CH), 37, 20, W, H)[:H, :W])
    assert np.all(ta.download().view(np.uint32) == 0x7f7f7f7f)  # the temp plane is not touched on this route


@pytest.mark.parametrize("constancy", [0, 1])
@pytest.mark.parametrize("outer,inner", [(2, 3), (3, 5)])
def test_solve_operator_swaps_callers_pointers(flow2d, oracle, rig, constancy, outer, inner):
    """dev_flow_du/dv and dev_temp_du/dv are taken BY POINTER: after Execute the caller's dev_flow_du / dev_flow_dv
    variables designate the planes holding the result (cuda_operation_solve_2d.cpp:128-142,288-289)."""
    up, make = rig
    f0, f1, u, v, _, _ = level_fields(oracle, W, H, 23)
    size = flow2d.DataSize3(W, H, 0)
    p = {n: up(n, a) for n, a in (("f0", f0), ("f1", f1), ("u", u), ("v", v))}
    du, dv, tdu, tdv = dp(up("du2")), dp(up("dv2")), dp(up("tdu2")), dp(up("tdv2"))
    op = make("solve", constancy=constancy)
    planes = {"du": du.value, "dv": dv.value, "tdu": tdu.value, "tdv": tdv.value}
    hx, hy = np.float32(CW / W), np.float32(CH / H)
    op.execute(dev_frame_0=dp(p["f0"]), dev_frame_1=dp(p["f1"]), dev_flow_u=dp(p["u"]), dev_flow_v=dp(p["v"]),
               dev_flow_du=du, dev_flow_dv=dv, dev_phi=dp(up("phi2")), dev_ksi=dp(up("ksi2")), dev_temp_du=tdu,
               dev_temp_dv=tdv, data_constancy=C.c_int(constancy), outer_iterations_count=C.c_size_t(outer),
               inner_iterations_count=C.c_size_t(inner), equation_alpha=C.c_float(3.5),
               equation_smoothness=C.c_float(0.001), equation_data=C.c_float(0.001), data_size=size,
               hx=C.c_float(hx), hy=C.c_float(hy), solver_algorithm=C.c_int(1))
    # per-sweep algorithm: outer*inner swaps, like the reference
    swapped = (outer * inner) % 2 == 1
    assert du.value == (planes["tdu"] if swapped else planes["du"])
    assert {du.value, tdu.value} == {planes["du"], planes["tdu"]} and {dv.value, tdv.value} == {planes["dv"], planes["tdv"]}
    odu, odv, _, _ = oracle.solve_level(f0, f1, u, v, W, H, hx, hy, 3.5, 0.001, 0.001, outer, inner, constancy)

    class View:  # download through a raw pointer
        def __init__(self, ptr, like):
            self.ptr, self.pitch, self.ctx, self.width, self.height = ptr, like.pitch, like.ctx, like.width, like.height
    got_du = flow2d.Plane.download(View(du.value, p["f0"]), W, H)
    got_dv = flow2d.Plane.download(View(dv.value, p["f0"]), W, H)
    assert np.array_equal(got_du, odu) and np.array_equal(got_dv, odv)
