"""Host-buffer entry (OpticalFlow2D::ComputeFlow: upload + pyramid + download) vs device-resident entry for one
workload: the PCIe-inclusive rate quoted in DESIGN.md (never bench.py's `value`)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

F = importlib.import_module("cuda-flow2d_amd")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
    cfg = bench.WORKLOADS[name]
    w, h = cfg["w"], cfg["h"]
    flow = F.OpticalFlow(w, h, cfg["constancy"])
    p = flow.params(cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001, cfg["median"],
                    cfg["sigma"])
    f0, f1 = bench.synthetic_pair(w, h, cfg["dx"], cfg["dy"])
    flow.compute_flow(f0, f1, p)
    reps, dev_ms = 3, 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        _, _, ms = flow.compute_flow(f0, f1, p)
        dev_ms += ms
    wall = (time.perf_counter() - t0) / reps * 1e3
    px_iters = w * h * cfg["outer"] * cfg["inner"]
    print("%s: ComputeFlow with host Data2D in/out: %.2f ms wall per pair (device events %.2f ms) -> %.1f pairs/s, "
          "%.0f Mpixel*iters/s PCIe-inclusive" % (name, wall, dev_ms / reps, 1e3 / wall, px_iters / wall / 1e3))
    flow.close()


if __name__ == "__main__":
    main()
