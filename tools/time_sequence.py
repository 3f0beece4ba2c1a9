"""Image sequence vs the same pairs computed one by one, one stream (developer tool).
usage: python tools/time_sequence.py [workload] [frames]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = importlib.import_module("cuda-flow2d_amd")
import bench  # noqa: E402


def main():
    cfg = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    w, h = cfg["w"], cfg["h"]
    ctx = F.Context(0)
    flow = F.OpticalFlow(w, h, cfg["constancy"], ctx=ctx)
    p = flow.params(cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001, cfg["median"],
                    cfg["sigma"])
    frames = [ctx.plane(w, h, bench.synthetic_pair(w, h, 0.7 * k, -0.4 * k)[1]) for k in range(n)]
    us = [ctx.plane(w, h) for _ in range(n - 1)]
    vs = [ctx.plane(w, h) for _ in range(n - 1)]

    def sequence():
        flow.compute_flow_sequence_device([f.ptr for f in frames], [u.ptr for u in us], [v.ptr for v in vs], p)

    def pairwise():
        for k in range(n - 1):
            flow.compute_flow_device(frames[k].ptr, frames[k + 1].ptr, us[k].ptr, vs[k].ptr, p)

    for name, fn in (("sequence", sequence), ("pair by pair", pairwise)):
        fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / 3 / (n - 1) * 1e3
        print("%-14s %.3f ms per flow (%d frames, one stream, eager launches)" % (name, ms, n))
    flow.close()
    ctx.close()


if __name__ == "__main__":
    main()
