"""Developer tool (round 6): one pair alone on the device, graph replay, for a kernel trace of its timeline.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/lone_pair_trace.py [workload] [lone 0|1] [replays]
tools/lone_pair_timeline.py OUT/*/*kernel_trace.csv prints the last replay kernel by kernel (queue, start, end, gap)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")
import bench  # noqa: E402  (the workloads' parameters)


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3_4096_gradient"
    lone = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    replays = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cfg = bench.WORKLOADS[workload]
    w, h = cfg["w"], cfg["h"]
    from oracle import oracle as O
    f0, f1 = O.synthetic_pair(w, h, 1.5, -0.75, seed=1, noise=True)
    c = F.Context(0)
    flow = F.OpticalFlow(w, h, cfg["constancy"], ctx=c, lone=lone)
    p = flow.params(cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001, cfg["median"], cfg["sigma"])
    a, b, u, v = c.plane(w, h, f0), c.plane(w, h, f1), c.plane(w, h), c.plane(w, h)
    flow.use_graph(True)
    for _ in range(replays):
        e0, e1 = c.event(), c.event()
        c.record(e0)
        flow.compute_flow_device(a.ptr, b.ptr, u.ptr, v.ptr, p, 0)
        c.record(e1)
        c.synchronize()
        print("pair: %.3f ms" % c.elapsed_ms(e0, e1))
    flow.close()
    c.close()


if __name__ == "__main__":
    main()
