"""Host-side cost of replaying one recorded pyramid (hipGraphLaunch of ~450 kernel nodes) vs launching it
eagerly (developer tool).  usage: python tools/graph_launch_cost.py [workload]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = importlib.import_module("cuda-flow2d_amd")
import bench  # noqa: E402


def main():
    cfg = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD]
    w, h = cfg["w"], cfg["h"]
    ctx = F.Context(0)
    flow = F.OpticalFlow(w, h, cfg["constancy"], ctx=ctx)
    params = flow.params(cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001,
                         cfg["median"], cfg["sigma"])
    f0, f1 = bench.synthetic_pair(w, h, cfg["dx"], cfg["dy"])
    planes = [ctx.plane(w, h, f0), ctx.plane(w, h, f1), ctx.plane(w, h), ctx.plane(w, h)]
    for graph in (True, False):
        flow.use_graph(graph)
        for _ in range(2):
            flow.compute_flow_device(*(p.ptr for p in planes), params, 0)
        ctx.synchronize()
        host, total = [], []
        for _ in range(6):
            t0 = time.perf_counter()
            flow.compute_flow_device(*(p.ptr for p in planes), params, 0)
            t1 = time.perf_counter()
            ctx.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) * 1e3)
            total.append((t2 - t0) * 1e3)
        print("%s: host call %.3f ms (min %.3f), until the GPU is done %.3f ms" %
              ("graph replay" if graph else "eager", float(np.median(host)), min(host), float(np.median(total))))
    flow.close()
    ctx.close()


if __name__ == "__main__":
    main()
