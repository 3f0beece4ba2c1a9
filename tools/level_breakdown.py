"""Per-level solve time vs whole-pyramid time of one workload (developer tool)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

F = importlib.import_module("cuda-flow2d_amd")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
    algo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cfg = bench.WORKLOADS[name]
    w, h = cfg["w"], cfg["h"]
    ctx = F.Context(0)
    flow = F.OpticalFlow(w, h, cfg["constancy"], ctx=ctx)
    p = flow.params(cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001, cfg["median"],
                    cfg["sigma"], algo)
    f0, f1 = bench.synthetic_pair(w, h, cfg["dx"], cfg["dy"])
    planes = [ctx.plane(w, h, f0), ctx.plane(w, h, f1), ctx.plane(w, h), ctx.plane(w, h)]
    ptrs = [pl.ptr for pl in planes]
    for _ in range(2):
        flow.compute_flow_device(*ptrs, p, 0)
    ctx.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        flow.compute_flow_device(*ptrs, p, 0)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    flow.reset_timings()
    flow.compute_flow_device(*ptrs, p, 1)
    ctx.synchronize()
    recs = flow.level_timings()
    total_solve = sum(r[2] for r in recs)
    print("%s algo %d: pyramid wall %.3f ms/pair, sum of level solves %.3f ms, rest %.3f ms" %
          (name, algo, wall, total_solve, wall - total_solve))
    for r in recs:
        print("   level %5dx%-5d solve %.3f ms  (%d launches)" % (r[0], r[1], r[2], r[4]))
    flow.close()
    ctx.close()


if __name__ == "__main__":
    main()
