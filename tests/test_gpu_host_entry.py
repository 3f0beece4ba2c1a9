"""The H<->D-inclusive batch entry (OpticalFlowBatch2D::ComputeFlowBatch: host Data2D images in, host flows out, uploads
and downloads pipelined against the lanes' pyramids) against the oracle, pair by pair, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("lanes,constancy,sigma", [(3, 0, 1.5), (2, 1, 0.0)])
def test_host_entry_pipeline_matches_the_oracle(flow2d, oracle, pinned, lanes, constancy, sigma):
    """11 distinct pairs through `lanes` lanes: every lane's staging planes are reused several times, upload, pyramid and
    download of a pair queued on the lane's stream; graph replay and eager; a second call continues on the same lanes."""
    w, h, n = 208, 144, 11
    p = (4, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, sigma)
    pairs = [oracle.synthetic_pair(w, h, 1.0 + 0.3 * k, -0.25 * k, seed=70 + k, noise=True) for k in range(n)]
    want = [oracle.compute_flow(f0, f1, *p, constancy)[:2] for f0, f1 in pairs]
    batch = flow2d.OpticalFlowBatch(w, h, constancy, lanes=lanes)
    images = []
    try:
        f0s = [flow2d.HostImage(w, h, pinned, q[0]) for q in pairs]
        f1s = [flow2d.HostImage(w, h, pinned, q[1]) for q in pairs]
        us = [flow2d.HostImage(w, h, pinned) for _ in pairs]
        vs = [flow2d.HostImage(w, h, pinned) for _ in pairs]
        images = f0s + f1s + us + vs
        assert all(q.pinned == pinned for q in images)
        for graph in (True, False):
            batch.use_graph(graph)
            for q in us + vs:
                q.array[...] = -7.0
            batch.compute_flow_batch(f0s[:7], f1s[:7], us[:7], vs[:7], batch.params(*p))
            batch.compute_flow_batch(f0s[7:], f1s[7:], us[7:], vs[7:], batch.params(*p), first_lane=1)
            batch.synchronize()
            for k in range(n):
                assert np.array_equal(us[k].array, want[k][0]) and np.array_equal(vs[k].array, want[k][1]), (graph, k)
                assert np.array_equal(f0s[k].array, pairs[k][0])  # the frames are only read
    finally:
        batch.close()
        for q in images:
            q.close()


def test_host_entry_lock_step_groups(flow2d, oracle):
    """group_size 3: consecutive host pairs are uploaded one below the other into the group's tall staging planes and
    computed with one launch per kernel; a count that does not fill the groups is refused."""
    w, h, G, n = 101, 75, 3, 9
    p = (3, 0.5, 2, 4, 35.0, 0.001, 0.001, 5, 1.5)
    pairs = [oracle.synthetic_pair(w, h, 0.5 * k, 1.0 - 0.3 * k, seed=90 + k, noise=True) for k in range(n)]
    batch = flow2d.OpticalFlowBatch(w, h, flow2d.GREY, lanes=2, group_size=G)
    images = []
    try:
        f0s = [flow2d.HostImage(w, h, True, q[0]) for q in pairs]
        f1s = [flow2d.HostImage(w, h, True, q[1]) for q in pairs]
        us = [flow2d.HostImage(w, h, True) for _ in pairs]
        vs = [flow2d.HostImage(w, h, True) for _ in pairs]
        images = f0s + f1s + us + vs
        batch.compute_flow_batch(f0s, f1s, us, vs, batch.params(*p))
        batch.synchronize()
        for k, (f0, f1) in enumerate(pairs):
            ou, ov, _ = oracle.compute_flow(f0, f1, *p, flow2d.GREY)
            assert np.array_equal(us[k].array, ou) and np.array_equal(vs[k].array, ov), k
        with pytest.raises(flow2d.Flow2DError):
            batch.compute_flow_batch(f0s[:4], f1s[:4], us[:4], vs[:4], batch.params(*p))
    finally:
        batch.close()
        for q in images:
            q.close()


def test_host_entry_rejects_wrong_sizes(flow2d):
    batch = flow2d.OpticalFlowBatch(64, 48, flow2d.GREY, lanes=1)
    a, b = flow2d.HostImage(64, 48, False), flow2d.HostImage(32, 48, False)
    try:
        with pytest.raises(flow2d.Flow2DError):
            batch.compute_flow_batch([a], [a], [a], [b], batch.params(2, 0.5, 1, 1, 35.0, 0.001, 0.001, 5, 1.5))
    finally:
        batch.close()
        a.close()
        b.close()


def test_pinned_data2d_through_compute_flow(flow2d, oracle):
    """OpticalFlow2D::ComputeFlow with page-locked Data2D images (Data2D::UsePinnedMemory, what the CLI switches on):
    same bits as with pageable ones."""
    w, h = 160, 120
    f0, f1 = oracle.synthetic_pair(w, h, 1.5, -0.75, seed=3, noise=True)
    p = (3, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5)
    ou, ov, _ = oracle.compute_flow(f0, f1, *p)
    L = flow2d.host_lib()
    flow = flow2d.OpticalFlow(w, h)
    try:
        L.flow2d_host_use_pinned_memory(1)
        u, v, _ = flow.compute_flow(f0, f1, flow.params(*p))
        assert np.array_equal(u, ou) and np.array_equal(v, ov)
    finally:
        L.flow2d_host_use_pinned_memory(0)
        flow.close()
