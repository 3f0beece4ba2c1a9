#!/usr/bin/env python3
"""bench.py -- the measured hot path: OpticalFlow2D::ComputeFlowDevice (C++ host layer -> C-ABI -> HIP
kernels for gfx950) on synthetic translating-sinusoid pairs, with the roofline of the dominant solver
kernel and the CPU oracle timed beside it.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" = one full coarse-to-fine run over one image pair per rank (inputs already resident in HBM).
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank works on its own pairs,
no data-path collective (independent pairs: SURVEY 8e) -> weak scaling.  Rank 0 prints ONE JSON line.

metric  = Mpixel*solver-iterations/s at the finest level (BASELINE.json): finest-level pixel-iterations
          (W*H*outer*inner per pair) of all ranks divided by the whole-pyramid wall time of the timed
          region, i.e. a whole-job rate (the pure finest-level solve rate is reported in
          "finest_level").  pairs_per_s is the second half of BASELINE.json's metric.
roofline: the finest level's dominant solver kernel (Jacobi sweep, or the fused outer-iteration kernel):
          achieved = algorithmic bytes per launch (40 B per pixel-sweep, 32 B per pixel for phi/ksi;
          SURVEY 8d) / average launch duration measured with HIP events on the launch stream inside
          the timed steps; peak = 8 TB/s HBM3E.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

# BASELINE.json configs[1..4] with the concrete parameters of SURVEY.md 8(d)
WORKLOADS = {
    # configs[2]: the HBM-roofline run the north_star's target is quoted on
    "cfg3_4096_gradient": dict(w=4096, h=4096, dx=2.0, dy=1.0, seed=3, constancy=1, levels=8, scale=0.5, outer=10,
                               inner=5, median=5, sigma=1.5, alpha=35.0, pairs_per_rank=1),
    "cfg3_4096_grey": dict(w=4096, h=4096, dx=2.0, dy=1.0, seed=3, constancy=0, levels=8, scale=0.5, outer=10,
                           inner=5, median=5, sigma=1.5, alpha=35.0, pairs_per_rank=1),
    # configs[1]
    "cfg2_1024_grey": dict(w=1024, h=1024, dx=1.5, dy=-0.75, seed=1, constancy=0, levels=5, scale=0.5, outer=10,
                           inner=5, median=5, sigma=1.5, alpha=35.0, pairs_per_rank=1),
    # configs[3]: 1920x1080 pairs, 8 per GPU
    "cfg4_1080p_batch": dict(w=1920, h=1080, dx=2.0, dy=0.0, seed=0, constancy=0, levels=8, scale=0.5, outer=10,
                             inner=5, median=5, sigma=1.5, alpha=35.0, pairs_per_rank=8),
    # configs[4]: 8192^2 large-displacement pair, all 12 levels
    "cfg5_8192_grey": dict(w=8192, h=8192, dx=12.0, dy=-7.0, seed=5, constancy=0, levels=12, scale=0.5, outer=10,
                           inner=5, median=5, sigma=1.5, alpha=35.0, pairs_per_rank=1),
}
DEFAULT_WORKLOAD = "cfg3_4096_gradient"


def synthetic_pair(w, h, dx, dy):
    """SURVEY 8(d) generator: I0 = 128 + 60 sin(2 pi x/64) cos(2 pi y/48) + 30 sin(2 pi (x+2y)/23.7),
    I1(x,y) = I0(x-dx, y-dy), evaluated in double, stored float32 (noise off)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)

    def img(xx, yy):
        return (128.0 + 60.0 * np.sin(2 * np.pi * xx / 64.0) * np.cos(2 * np.pi * yy / 48.0)
                + 30.0 * np.sin(2 * np.pi * (xx + 2 * yy) / 23.7))

    return img(x, y).astype(np.float32), img(x - dx, y - dy).astype(np.float32)


def cpu_baseline(cfg, budget_s=20.0):
    """The CPU oracle (oracle/flow2d_oracle.c, OpenMP) timed on this box's host cores on a BOUNDED sample of
    the same workload: the same pyramid/solver parameters on a centre crop sized so the run stays within
    ~budget_s seconds of CPU work.  Same metric definition as `value`.  Rank 0, N = 1 only."""
    from oracle import oracle as O

    O.lib()
    threads = O.max_threads()
    # calibrate on a small crop, then pick the largest power-of-two crop inside the CPU-seconds budget
    probe = 256
    f0, f1 = synthetic_pair(probe, probe, cfg["dx"], cfg["dy"])
    t0 = time.perf_counter()
    O.compute_flow(f0, f1, cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001,
                   cfg["median"], cfg["sigma"], cfg["constancy"])
    t_probe = time.perf_counter() - t0
    per_px_cpu_s = t_probe * threads / (probe * probe)
    side = probe
    while side * 2 <= min(cfg["w"], cfg["h"]) and per_px_cpu_s * (side * 2) ** 2 <= budget_s:
        side *= 2
    f0, f1 = synthetic_pair(side, side, cfg["dx"], cfg["dy"])
    t0 = time.perf_counter()
    _, _, t_finest = O.compute_flow(f0, f1, cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"],
                                    0.001, 0.001, cfg["median"], cfg["sigma"], cfg["constancy"])
    t = time.perf_counter() - t0
    px_iters = side * side * cfg["outer"] * cfg["inner"]
    # the same code on ONE thread (SURVEY 8d asks for both), on the small calibration crop
    O.set_threads(1)
    f0s, f1s = synthetic_pair(probe, probe, cfg["dx"], cfg["dy"])
    t0 = time.perf_counter()
    O.compute_flow(f0s, f1s, cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001,
                   cfg["median"], cfg["sigma"], cfg["constancy"])
    t_single = time.perf_counter() - t0
    O.set_threads(threads)
    return {
        "value": round(px_iters / t / 1e6, 2),
        "unit": "Mpixel*iters/s",
        "cores": threads,
        "kind": "port",
        "sample": "%dx%d crop of the workload's synthetic pair, same levels/outer/inner/constancy, one full "
                  "pyramid, %.2f s wall on %d OpenMP threads" % (side, side, t, threads),
        "finest_level_solve_mpix_iters_per_s": round(px_iters / t_finest / 1e6, 2) if t_finest > 0 else None,
        "single_thread": {"value": round(probe * probe * cfg["outer"] * cfg["inner"] / t_single / 1e6, 2),
                          "sample": "%dx%d crop, %.2f s wall on 1 thread" % (probe, probe, t_single)},
    }


def load_traffic(workload, algorithm):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/traffic.json, written by tools/pmc_traffic.py); None if not measured for this workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None
    entry = table.get("%s/algorithm%d" % (workload, algorithm))
    return entry.get("hbm_bytes_per_launch") if entry else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--algorithm", type=int, default=0, help="flow2d_solver_algorithm: 0 auto, 1 per-sweep, 2 fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying HIP graphs")
    ap.add_argument("--pipeline", type=int, default=4,
                    help="independent pairs in flight per GPU when a step holds a single pair: consecutive steps go to "
                         "alternating streams so one pair's launch-bound coarse levels overlap the next pair's fine levels")
    args = ap.parse_args()
    cfg = WORKLOADS[args.workload]

    flow2d = importlib.import_module("cuda-flow2d_amd")
    batch = importlib.import_module("cuda-flow2d_amd.batch")
    if not (os.path.exists(flow2d.HIP_LIB_PATH) and os.path.exists(flow2d.HOST_LIB_PATH)):
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            flow2d.build()  # checkout without the in-tree libraries (normally built by __graft_entry__.build())
        else:
            for _ in range(600):  # the other ranks wait for rank 0's build
                if os.path.exists(flow2d.HIP_LIB_PATH) and os.path.exists(flow2d.HOST_LIB_PATH):
                    break
                time.sleep(0.5)
            time.sleep(1.0)
    rank, local_rank, world = batch.world_info()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" %
                     (args.gpus, world))
        args.gpus = world

    import torch  # device plumbing only: barrier, device-wide synchronise, max-over-ranks

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the flow2d path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    batch.init(backend="nccl", device=torch.device("cuda", local_rank))  # RCCL; no-op for one process

    # "lanes": independent (stream, OpticalFlow2D, plane pool) triples.  A rank with several pairs per step
    # spreads them over up to 4 lanes so the launch-bound coarse levels of one pair overlap another pair's work.
    w, h = cfg["w"], cfg["h"]
    n_lanes = max(1, min(4, cfg["pairs_per_rank"] * args.pipeline))
    lanes = []
    for _ in range(n_lanes):
        c = flow2d.Context(local_rank)
        f = flow2d.OpticalFlow(w, h, cfg["constancy"], ctx=c)
        lanes.append({"ctx": c, "flow": f, "pairs": []})
    ctx, flow = lanes[0]["ctx"], lanes[0]["flow"]
    # rank 0's parameter block on every rank (RCCL broadcast), then the same solve everywhere
    block = batch.broadcast_params([cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001,
                                    0.001, cfg["median"], cfg["sigma"], args.algorithm])
    params = flow.params(int(block[0]), block[1], int(block[2]), int(block[3]), block[4], block[5], block[6],
                         int(block[7]), block[8], int(block[9]))

    # this rank's pairs, resident in HBM before the timed region
    total_pairs = cfg["pairs_per_rank"] * world
    for n, gk in enumerate(batch.pairs_of_rank(total_pairs, rank, world)):  # pair k -> rank k mod world
        if args.workload == "cfg4_1080p_batch":
            dx, dy = 2.0 * np.cos(gk), 2.0 * np.sin(gk)
        else:
            dx, dy = cfg["dx"], cfg["dy"]
        f0, f1 = synthetic_pair(w, h, dx, dy)
        targets = [lanes[n % n_lanes]] if cfg["pairs_per_rank"] > 1 else lanes  # single pair: every lane has a copy
        for lane in targets:
            c = lane["ctx"]
            lane["pairs"].append((c.plane(w, h, f0), c.plane(w, h, f1), c.plane(w, h), c.plane(w, h)))
    free_b, total_b = ctx.mem_info()
    single = cfg["pairs_per_rank"] == 1

    def barrier():
        batch.barrier()
        for lane in lanes:
            lane["ctx"].synchronize()
        torch.cuda.synchronize()

    def step(index, instrumented):
        """One pass over this rank's pairs.  Replayed from recorded HIP graphs, except the instrumented pass,
        which launches eagerly with events around every level's solve and every finest-level solver launch.
        With one pair per step, step `index` goes to stream index mod n_lanes, so consecutive steps overlap."""
        active = [lanes[index % n_lanes]] if single else lanes
        if instrumented:  # the instrumented pass runs alone on lane 0 so its launch durations are undisturbed
            for lane in lanes:
                lane["ctx"].synchronize()
            active = [lanes[0]] if single else lanes
        for lane in active:
            lane["flow"].use_graph(not instrumented and not args.no_graph)
        for k in range(max(len(l["pairs"]) for l in active)):
            for lane in active:
                if k < len(lane["pairs"]):
                    pf0, pf1, pu, pv = lane["pairs"][k]
                    lane["flow"].compute_flow_device(pf0.ptr, pf1.ptr, pu.ptr, pv.ptr, params,
                                                     2 if instrumented else 0)

    for k in range(max(args.warmup, 1) * n_lanes):
        step(k, False)  # also records the graphs of every lane
    if args.warmup > 0:
        step(0, True)
    flow.reset_timings()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, k == args.steps - 1)  # the last timed step is the instrumented one (roofline sample)
    barrier()
    elapsed = time.perf_counter() - t0
    finest = [r for r in flow.level_timings() if (r[0], r[1]) == (w, h)]
    elapsed = batch.max_over_ranks(elapsed)

    if rank == 0:
        pairs_total = args.steps * cfg["pairs_per_rank"] * world
        px_iters = float(w) * h * cfg["outer"] * cfg["inner"]
        solve_ms = float(np.mean([r[2] for r in finest]))
        kernel_ms = float(np.mean([r[3] / r[4] for r in finest]))
        launches = finest[-1][4]
        bytes_per_launch = float(finest[-1][5])
        algorithm_used = 2 if launches == cfg["outer"] and cfg["inner"] > 1 else 1
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mpixels*SOR-iters/sec at finest level (whole-pyramid wall time); full-pyramid pairs/sec in pairs_per_s",
            "value": round(px_iters * pairs_total / elapsed / 1e6, 1),
            "unit": "Mpixel*iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload, "width": w, "height": h, "pairs_per_gpu_per_step": cfg["pairs_per_rank"],
                "pyramid_levels": int(min(cfg["levels"], flow.max_warp_level(w, h, cfg["scale"]))),
                "warp_scale": cfg["scale"], "outer_iterations": cfg["outer"], "inner_iterations": cfg["inner"],
                "data_constancy": "gradient" if cfg["constancy"] else "grey", "median_radius": cfg["median"],
                "gaussian_sigma": cfg["sigma"], "alpha": cfg["alpha"], "solver_algorithm": algorithm_used,
                "relaxation": "Jacobi, reference iteration counts (bit-exact parity mode)",
                "parallelism": "independent pairs, one process per GPU, no data-path collective",
                "streams_per_gpu": n_lanes, "hip_graph_replay": not args.no_graph,
            },
            "pairs_per_s": round(pairs_total / elapsed, 3),
            "finest_level": {
                "solve_ms": round(solve_ms, 4),
                "mpix_iters_per_s": round(px_iters / (solve_ms * 1e-3) / 1e6, 1),
                "algorithmic_gbs": round(w * h * cfg["outer"] * (32 + 40 * cfg["inner"]) / (solve_ms * 1e-3) / 1e9, 1),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("fused outer-iteration kernel (phi/ksi + %d Jacobi sweeps)" % cfg["inner"])
                if algorithm_used == 2 else "Jacobi sweep kernel (solve_2d%s)" % ("_grad" if cfg["constancy"] else ""),
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": load_traffic(args.workload, algorithm_used),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": round(kernel_ms, 5),
                "launches_per_level_solve": launches,
            },
            "device_memory": {"used_gib": round((total_b - free_b) / 2 ** 30, 3), "total_gib": round(total_b / 2 ** 30, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))

    for lane in lanes:
        lane["flow"].close()
        lane["ctx"].close()
    batch.shutdown()


if __name__ == "__main__":
    main()
