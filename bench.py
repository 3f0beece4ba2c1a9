#!/usr/bin/env python3
"""bench.py -- the measured hot path: OpticalFlowBatch2D / OpticalFlow2D::ComputeFlowDevice (C++ host layer -> C-ABI -> HIP
kernels for gfx950) on synthetic translating-sinusoid pairs, with the roofline of the dominant solver
kernel, the CPU oracle and (N = 1) the reference's own kernels timed beside it.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" = one full coarse-to-fine run over one image pair per rank (inputs already resident in HBM).
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank works on its own pairs,
no data-path collective (independent pairs: SURVEY 8e) -> weak scaling.  Rank 0 prints ONE JSON line.

Order of a run
  0. batch leg (BASELINE.json configs[3]), a side figure of the line (`batch`): 8 pairs of 1920x1080 per GPU as one lock-step group
     per step on 4 streams, >= 64 steps, median of three regions; then the flow fields of all ranks are gathered to rank 0 over RCCL
     (timed separately: gather.ms).  It runs first, on an untouched GPU, and is closed before the main job allocates (behind the
     main job it read low by amounts that depended on what had run before it)
  1. warm-up of the main job (records the HIP graphs), barrier
  2. TIMED REGIONS: `--repeats` (5) regions of exactly K steps each, replayed from the recorded graphs, nothing else; barrier
     around each; max over ranks; the line reports the MEDIAN region (ms_per_step, value, pairs_per_s) and the fastest / slowest
  3. output check: the flow fields the timed steps left in HBM are hashed; an eager (un-graphed) recomputation
     must give the same bits, and the first pair's flow must equal the CPU oracle's in every pixel (output_check.oracle);
     every plane set in flight holds a DIFFERENT synthetic pair (distinct_pairs_in_flight)
  4. host-entry leg: the same pairs from HOST images to HOST flows (uploads and downloads inside the bracket, pipelined
     against the pyramids by OpticalFlowBatch2D::ComputeFlowBatch): pairs_per_s_incl_h2d, SURVEY 8(d) metric 2 as defined
  5. roofline sample: eager passes with HIP events on the launch stream around every finest-level solver launch; the shader clock
     the chip HOLDS under the finest level's solves (one sleeping wave per XCD, flow2d_clock_probe_*); a lone pair's latency (a lone
     object and a pipeline lane's taking turns); the per-sweep kernel alone; a device-to-device copy for scale; the strip kernel's
     timing probes (developer libraries under ab/) when present
  6. baselines on rank 0 at N = 1: the CPU oracle on the workload's pair; the reference's own kernels (oracle/_ref,
     compiled from its sources for gfx950) with the reference's launch schedule on this GPU
  (--workload cfg3_4096_sor: the opt-in red-black SOR mode and its leg against Jacobi on the whole pyramid)

metric  = Mpixel*solver-iterations/s at the finest level (BASELINE.json): finest-level pixel-iterations
          (W*H*outer*inner per pair) of all ranks / whole-pyramid wall time of the timed region (a whole-job
          rate; the pure finest-level solve rate is in "finest_level").  pairs_per_s: the metric's second half.
roofline: the finest level's dominant solver kernel.  achieved/frac/traffic are PHYSICAL HBM bytes per launch (rocprofv3
          --pmc passes of this very workload, run as child processes before the timed region) over the launch duration
          measured with HIP events on the launch stream; effective_* are the contract's ALGORITHMIC figures (SURVEY 8d:
          40 B per pixel-sweep, 32 B per pixel for phi/ksi, i.e. the reference's per-sweep schedule) over the same
          duration -- above 1 for the fused kernel because it does not move those bytes.  bound says what limits the
          kernel: "valu" (vector-instruction issue, valu_issue_frac) for the temporally blocked kernels, "hbm" otherwise.
          Round 6 adds the launch's distance to each bound: shader_clock_ghz (+ per XCD; the power management holds 1.6-2.1 GHz under
          the strip kernel, not the nominal 2.4) and valu_issue_frac_at_clock, hbm_floor_us (compulsory bytes / this box's copy rate),
          memory_only_us / compute_only_us (the probes; each at the clock IT holds).
"""
import argparse
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; RCCL's own stream
# (created by the process group before the four lanes exist) would make two lanes share a queue: 205 instead of 224
# pairs/s under torch.distributed.run.  Eight queues leave every lane its own.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS = 256 * 4            # CUs x SIMDs per CU
CLOCK_HZ = 2.4e9           # max shader clock
VALU_ISSUE_CYCLES = 2.0    # a wave64 VALU instruction occupies a SIMD-32 for 2 cycles (same guide)

# BASELINE.json configs[0..4] with the concrete parameters of SURVEY.md 8(d)
WORKLOADS = {
    # configs[0]: the reference's own sample pair with the values of its settings.xml (settings.xml:17-19: 20 levels at 0.9,
    # 20 x 5 sweeps, alpha 3.5, sigma 0.45, median 5): the launch-bound shape -- 20 levels of at most 584 x 388 pixels
    "cfg1_rub": dict(w=584, h=388, dx=0.0, dy=0.0, seed=0, constancy=0, levels=20, scale=0.9, outer=20, inner=5, median=5,
                     sigma=0.45, alpha=3.5, pairs_per_rank=1, frames="rub"),
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
# The north_star names "red-black SOR relaxation"; the reference is Jacobi, so SOR is an opt-in mode without
# reference parity (checked against its own oracle restatement).  This workload is config 3 with every inner iteration a
# red-black SOR iteration, temporally blocked in the strip kernel (two iterations per launch), plus the time-to-residual
# leg: how many SOR iterations per outer iteration -- and how many ms -- reach the residual Jacobi 10 x 5 reaches.
WORKLOADS["cfg3_4096_sor"] = dict(WORKLOADS["cfg3_4096_gradient"], sor_omega=1.9, inner=1)  # (what the leg finds fastest to Jacobi's error)
# developer what-if (not a BASELINE config): 4 / 8 pairs of config 4 stacked into one tall frame, i.e. the kernel sizes a
# lock-step batch of pairs would launch
WORKLOADS["x_stack4_1080p"] = dict(WORKLOADS["cfg4_1080p_batch"], h=4320, pairs_per_rank=2)
WORKLOADS["x_stack8_1080p"] = dict(WORKLOADS["cfg4_1080p_batch"], h=8640, pairs_per_rank=1)
WORKLOADS["x_stack4_1024"] = dict(WORKLOADS["cfg2_1024_grey"], h=4096, pairs_per_rank=1)
DEFAULT_WORKLOAD = "cfg3_4096_gradient"
BATCH_WORKLOAD = "cfg4_1080p_batch"
CONSTANCY_NAME = {0: "grey", 1: "gradient", 2: "gradient-untiled", 3: "log-derivatives"}


def synthetic_pair(w, h, dx, dy):
    """SURVEY 8(d) generator: I0 = 128 + 60 sin(2 pi x/64) cos(2 pi y/48) + 30 sin(2 pi (x+2y)/23.7),
    I1(x,y) = I0(x-dx, y-dy), evaluated in double, stored float32 (noise off)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)

    def img(xx, yy):
        return (128.0 + 60.0 * np.sin(2 * np.pi * xx / 64.0) * np.cos(2 * np.pi * yy / 48.0)
                + 30.0 * np.sin(2 * np.pi * (xx + 2 * yy) / 23.7))

    return img(x, y).astype(np.float32), img(x - dx, y - dy).astype(np.float32)


def workload_pair(workload, cfg, k, world=1):
    """Global pair k of a workload: the SURVEY 8(d) synthetic pair, or (cfg1_rub) the reference's sample frames rub1 / rub2
    (tests/data, 584 x 388 u8 raws widened to float like Data2D::ReadRAWFromFileU8 does; the further pairs a rank keeps in
    flight are the same frames rolled by j rows and columns, so that every plane set holds data of its own)."""
    if cfg.get("frames") == "rub":
        d = os.path.join(ROOT, "tests", "data")
        j = k // max(world, 1)
        return tuple(np.roll(np.fromfile(os.path.join(d, n), np.uint8).reshape(cfg["h"], cfg["w"]).astype(np.float32),
                             (j, 2 * j), axis=(0, 1)) for n in ("rub1.raw", "rub2.raw"))
    return synthetic_pair(cfg["w"], cfg["h"], *pair_shift(workload, cfg, k, world))


def workload_pairs(workload, cfg, ks, world=1):
    """Several pairs at once, generated side by side (numpy releases the GIL in its transcendental loops)."""
    from concurrent.futures import ThreadPoolExecutor

    if len(ks) <= 1:
        return [workload_pair(workload, cfg, k, world) for k in ks]
    with ThreadPoolExecutor(max_workers=min(8, len(ks))) as pool:
        return list(pool.map(lambda k: workload_pair(workload, cfg, k, world), ks))


def pair_shift(workload, cfg, k, world=1):
    """Displacement of global pair k.  Config 4: SURVEY 8d's (2 cos k, 2 sin k).  The single-pair workloads keep a rank's
    first pair (k < world) at the workload's own (dx, dy) -- the pair the CPU oracle checks -- and move the further pairs a
    rank keeps in flight (k = rank + world * j: plane set j of the rank) by j quarter / eighth pixels more."""
    if workload == BATCH_WORKLOAD:
        return 2.0 * np.cos(k), 2.0 * np.sin(k)
    j = k // max(world, 1)
    return cfg["dx"] + 0.25 * j, cfg["dy"] - 0.125 * j


def cpu_baseline(cfg, budget_s=20.0, full_run=None):
    """The CPU oracle (oracle/flow2d_oracle.c, OpenMP) timed on this box's host cores, same metric definition as `value`.
    Sample: the WHOLE workload pair when the output check has just run the oracle on it (full_run = (wall seconds,
    finest-level solve seconds): at 4096^2 about 5 s on 16 threads), otherwise the same pyramid / solver parameters on
    the largest power-of-two centre crop that stays within ~budget_s seconds of CPU work.  Rank 0, N = 1 only."""
    from oracle import oracle as O

    O.lib()
    threads = O.max_threads()
    probe = 256
    f0, f1 = synthetic_pair(probe, probe, cfg["dx"], cfg["dy"])
    t0 = time.perf_counter()
    O.compute_flow(f0, f1, cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001,
                   cfg["median"], cfg["sigma"], cfg["constancy"], sor_omega=cfg.get("sor_omega", 0.0))
    t_probe = time.perf_counter() - t0
    if full_run is not None:
        side_w, side_h, (t, t_finest) = cfg["w"], cfg["h"], full_run
        what = "the workload's whole %dx%d pair" % (side_w, side_h)
    else:
        # calibrate on the small crop, then pick the largest power-of-two crop inside the CPU-seconds budget
        per_px_cpu_s = t_probe * threads / (probe * probe)
        side = probe
        while side * 2 <= min(cfg["w"], cfg["h"]) and per_px_cpu_s * (side * 2) ** 2 <= budget_s:
            side *= 2
        f0, f1 = synthetic_pair(side, side, cfg["dx"], cfg["dy"])
        t0 = time.perf_counter()
        _, _, t_finest = O.compute_flow(f0, f1, cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"],
                                        0.001, 0.001, cfg["median"], cfg["sigma"], cfg["constancy"],
                                        sor_omega=cfg.get("sor_omega", 0.0))
        t = time.perf_counter() - t0
        side_w = side_h = side
        what = "%dx%d crop of the workload's synthetic pair" % (side, side)
    px_iters = side_w * side_h * cfg["outer"] * cfg["inner"]
    # the same code on ONE thread (SURVEY 8d asks for both), on the small calibration crop
    O.set_threads(1)
    f0s, f1s = synthetic_pair(probe, probe, cfg["dx"], cfg["dy"])
    t0 = time.perf_counter()
    O.compute_flow(f0s, f1s, cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001,
                   cfg["median"], cfg["sigma"], cfg["constancy"], sor_omega=cfg.get("sor_omega", 0.0))
    t_single = time.perf_counter() - t0
    O.set_threads(threads)
    return {
        "value": round(px_iters / t / 1e6, 2),
        "unit": "Mpixel*iters/s",
        "cores": threads,
        "kind": "port",
        "sample": "%s, same levels/outer/inner/constancy, one full pyramid, %.2f s wall on %d OpenMP threads" %
                  (what, t, threads),
        "pairs_per_s": round(1.0 / t, 4) if full_run is not None else None,
        "finest_level_solve_mpix_iters_per_s": round(px_iters / t_finest / 1e6, 2) if t_finest > 0 else None,
        "single_thread": {"value": round(probe * probe * cfg["outer"] * cfg["inner"] / t_single / 1e6, 2),
                          "sample": "%dx%d crop, %.2f s wall on 1 thread" % (probe, probe, t_single)},
    }


def reference_gpu_baseline(cfg, f0, f1):
    """THE REFERENCE'S OWN KERNELS (src/kernels/*_2d.cu compiled for gfx950 from its sources: oracle/_ref) with the
    reference's launch schedule -- one launch per sweep, a stream synchronisation after each, one pair at a time
    (cuda_operation_solve_2d.cpp:263-299) -- on this GPU, whole ComputeFlow incl. the H<->D copies the reference's
    own timer brackets (optical_flow_2d.cpp:173-179,548-554).  A checker-side baseline like cpu_baseline: runs after
    the timed region, on rank 0 at N = 1; None when oracle/_ref was not built or the mode has no reference kernel."""
    try:
        from oracle import ref_kernels as RK
    except Exception:  # noqa: BLE001 - the checker is optional here
        return None
    ref_mode = {0: 0, 1: 1, 3: 2}.get(cfg["constancy"])
    if not RK.available() or ref_mode is None:
        return None
    try:
        with RK.RefKernels(cfg["w"], cfg["h"]) as R:
            best = None
            for _ in range(2):
                _, _, total_ms, finest_ms = R.compute_flow(f0, f1, cfg["levels"], cfg["scale"], cfg["outer"],
                                                           cfg["inner"], cfg["alpha"], 0.001, 0.001, cfg["median"],
                                                           cfg["sigma"], constancy=ref_mode)
                if best is None or total_ms < best[0]:
                    best = (total_ms, finest_ms)
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:200]}
    px_iters = float(cfg["w"]) * cfg["h"] * cfg["outer"] * cfg["inner"]
    return {
        "what": "the reference's kernels (hipcc -ffp-contract=off build of its .cu files) under the reference's "
                "schedule: per-sweep launches, host sync after every sweep, one pair at a time, H<->D inside the bracket",
        "value": round(px_iters / (best[0] * 1e-3) / 1e6, 1),
        "unit": "Mpixel*iters/s",
        "pairs_per_s": round(1e3 / best[0], 3),
        "ms_per_pair": round(best[0], 3),
        "finest_level_solve_ms": round(best[1], 3),
        "finest_level_mpix_iters_per_s": round(px_iters / (best[1] * 1e-3) / 1e6, 1),
    }


def measured_copy_peak(flow2d, local_rank):
    """Device-to-device copy of 512 MiB (hipMemcpyAsync D2D, read + write counted) on a stream of its own, best of 5:
    the HBM rate this box reaches on a plain stream, next to the 8 TB/s nominal peak."""
    c = flow2d.Context(local_rank)
    try:
        n = 8192
        a, b = c.plane(n, 2 * n).fill_bytes(1), c.plane(n, 2 * n)
        nbytes = a.pitch * 2 * n
        best = None
        for _ in range(6):
            e0, e1 = c.event(), c.event()
            c.record(e0)
            if flow2d.hip_lib().flow2d_copy_d2d(c.handle, b.ptr, a.ptr, nbytes) != 0:
                return None
            c.record(e1)
            ms = c.elapsed_ms(e0, e1)
            best = ms if best is None or ms < best else best
        return round(2.0 * nbytes / (best * 1e-3) / 1e9, 1)
    except Exception:  # noqa: BLE001 - informational figure only
        return None
    finally:
        c.close()


PMC_CALIBRATION_ADDS = 4
PMC_EXTRA_SWEEPS = 6


def pmc_child(flow2d, args, cfg):
    """--pmc-child: what the rocprofv3 --pmc passes profile -- the workload's first pair, two eager pyramids on one stream
    (same data, same parameters, same AUTO choice as the timed run), nothing else."""
    w, h = cfg["w"], cfg["h"]
    c = flow2d.Context(0)
    flow = flow2d.OpticalFlow(w, h, cfg["constancy"], ctx=c)
    try:
        f0, f1 = (c.plane(w, h, a) for a in workload_pair(args.workload, cfg, 0))
        u, v = c.plane(w, h), c.plane(w, h)
        p = flow.params(cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001, 0.001,
                        cfg["median"], cfg["sigma"], args.algorithm, sor_omega=cfg.get("sor_omega", 0.0))
        for _ in range(2):
            flow.compute_flow_device(f0.ptr, f1.ptr, u.ptr, v.ptr, p, 0)
        c.synchronize()
        # after the pyramids: PMC_CALIBRATION_ADDS launches of add_2d at full resolution, whose traffic is known exactly
        # (2 planes read, 1 written: the FETCH_SIZE / WRITE_SIZE corrections), and PMC_EXTRA_SWEEPS launches of the per-sweep
        # Jacobi kernel of this data term at the finest level (roofline.per_sweep: its physical bytes)
        du, dv, phi, ksi, tdu, tdv = (c.plane(w, h).fill_bytes(0) for _ in range(6))
        for _ in range(PMC_CALIBRATION_ADDS):
            c.add(tdu, f0, w, h)
        c.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, 1.0, 1.0, 0.001, 0.001, phi, ksi)
        for _ in range(PMC_EXTRA_SWEEPS):
            c.solve_sweep(f0, f1, u, v, du, dv, phi, ksi, w, h, 1.0, 1.0, cfg["alpha"], tdu, tdv, cfg["constancy"])
        c.synchronize()
    finally:
        flow.close()
        c.close()


def pmc_passes(args, cfg):
    """The HBM bytes and VALU instructions of the dominant kernel FOR THIS WORKLOAD, measured in this run: before this
    process touches the GPU, three child processes `rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py
    --pmc-child ...` (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots; no other
    trace domain is combined with --pmc).  Bytes = FETCH_SIZE x correction + WRITE_SIZE (KiB): on gfx950 FETCH_SIZE tallies
    128-byte requests as 64 (same guide, HBM section), so the correction is 2; it is re-measured here on the
    PMC_CALIBRATION_ADDS explicit full-resolution add_2d launches the child makes after its pyramids, whose traffic is
    known exactly (2 planes read, 1 written; the pyramid itself launches no add_2d since its u += du rides in the median).  Returns {} when
    rocprofv3 is missing or a pass fails (the line then falls back to profiles/traffic.json and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if not shutil.which("rocprofv3"):
        return {}
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "this process is itself being profiled: no nested rocprofv3 passes"}
    out_root = tempfile.mkdtemp(prefix="flow2d_pmc_", dir="/tmp")
    rows = {}
    try:
        for n, counters in enumerate((["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_WAVES"])):
            d = os.path.join(out_root, "pass%d" % n)
            cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc"] + counters + \
                  ["-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload,
                   "--algorithm", str(args.algorithm)]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": "rocprofv3 pass %d: rc %d %s" % (n, r.returncode, r.stderr[-200:])}
            for f in files:
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                    rows.setdefault(row["Counter_Name"], []).append(
                        (int(row["Dispatch_Id"]), name, int(row["Grid_Size"]), float(row["Counter_Value"])))
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
        return {"error": str(e)[:200]}
    finally:
        shutil.rmtree(out_root, ignore_errors=True)
    return {"rows": rows}


def pmc_select(collected, cfg, algorithm_used):
    """Per-launch figures of the finest level's solver kernel (the algorithm the timed run used there) out of the rows
    pmc_passes collected."""
    if "rows" not in collected:
        return collected
    rows = collected["rows"]
    family = {1: "sweep_", 2: "fused_outer_kernel", 3: "small_level_kernel", 4: "tile_outer_kernel"}[algorithm_used]

    per_launch_max = 2 if cfg.get("sor_omega") else 5  # red-black iterations are two half-sweep stages each: two per launch
    per_solve = {1: cfg["outer"] * cfg["inner"], 2: cfg["outer"] * -(-cfg["inner"] // per_launch_max), 3: 1, 4: cfg["outer"]}[algorithm_used]
    PYRAMIDS = 2  # pmc_child runs the pyramid twice

    def launches(counter, prefix, per_pyramid):
        """[value per launch, in dispatch order] of the FINEST level's launches of the kernels named prefix*: a pyramid
        goes coarse to fine, so they are the last `per_pyramid` launches of that family in each pyramid (Grid_Size alone
        cannot tell 5120 x 25 from 2560 x 50)."""
        sel = sorted(r for r in rows.get(counter, []) if r[1].startswith(prefix))
        if prefix == "sweep_":
            sel = sel[:-PMC_EXTRA_SWEEPS]  # pmc_child's explicit sweeps after the pyramids
        n = len(sel) // PYRAMIDS
        if n < per_pyramid or len(sel) != n * PYRAMIDS:
            return None, []
        picked = [r for k in range(PYRAMIDS) for r in sel[(k + 1) * n - per_pyramid:(k + 1) * n]]
        return picked[-1][1], [r[3] for r in picked]

    dominant, _ = launches("FETCH_SIZE", family, per_solve)
    if dominant is None:
        return {"error": "no %s* launches in the counter rows" % family}
    plane = float(cfg["w"]) * cfg["h"] * 4

    def extras(counter, prefix, count):
        """the explicit launches pmc_child makes after the pyramids: the last `count` of their family (the first one left out: cold)"""
        sel = sorted(r for r in rows.get(counter, []) if r[1].startswith(prefix))[-count:]
        return [r[3] for r in sel[1:]], (sel[-1][1] if sel else None)

    add_fetch, _ = extras("FETCH_SIZE", "add_2d_kernel", PMC_CALIBRATION_ADDS)
    add_write, _ = extras("WRITE_SIZE", "add_2d_kernel", PMC_CALIBRATION_ADDS)
    # small frames live in the caches: the calibration takes planes of 32 MiB and more (three of them per add_2d launch,
    # re-used launch after launch, then no longer sit in the XCDs' 4 MiB L2s; smaller frames take the guide's x 2)
    calibrate = bool(add_fetch and add_write) and plane >= 32 * 2 ** 20
    corr = 2 * plane / (np.mean(add_fetch) * 1024) if calibrate else 2.0
    wcorr = plane / (np.mean(add_write) * 1024) if calibrate else 1.0
    out = {"kernel": dominant, "fetch_size_correction": round(float(corr), 4), "write_size_correction": round(float(wcorr), 4),
           "correction_from": ("%d full-resolution add_2d launches after the pyramids (2 planes read, 1 written)" % (PMC_CALIBRATION_ADDS - 1))
           if calibrate else "MI355X_MICROARCH.md (FETCH_SIZE x 2)"}
    sweep_fetch, sweep_name = extras("FETCH_SIZE", "sweep_", PMC_EXTRA_SWEEPS)
    sweep_write, _ = extras("WRITE_SIZE", "sweep_", PMC_EXTRA_SWEEPS)
    sweep_valu, _ = extras("SQ_INSTS_VALU", "sweep_", PMC_EXTRA_SWEEPS)
    if sweep_fetch and sweep_write:
        out["per_sweep"] = {"kernel": sweep_name, "launches_sampled": len(sweep_fetch),
                            "hbm_read_bytes_per_launch": round(float(np.mean(sweep_fetch)) * 1024 * corr),
                            "hbm_write_bytes_per_launch": round(float(np.mean(sweep_write)) * 1024 * wcorr),
                            "valu_insts_per_launch": round(float(np.mean(sweep_valu))) if sweep_valu else None}
    split = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVES"):
        _, vals = launches(counter, family, per_solve)
        if not vals:
            return {"error": "no %s rows for %s" % (counter, dominant)}
        first = [x for i, x in enumerate(vals) if i % per_solve == 0]
        steady = [x for i, x in enumerate(vals) if i % per_solve != 0] or first
        split[counter] = (float(np.mean(first)), float(np.mean(steady)), len(vals))
    out["launches_sampled"] = split["FETCH_SIZE"][2]
    out["hbm_bytes_first_launch"] = round(split["FETCH_SIZE"][0] * 1024 * corr + split["WRITE_SIZE"][0] * 1024 * wcorr)
    out["hbm_read_bytes_per_launch"] = round(split["FETCH_SIZE"][1] * 1024 * corr)
    out["hbm_write_bytes_per_launch"] = round(split["WRITE_SIZE"][1] * 1024 * wcorr)
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    out["valu_insts_per_launch"] = round(split["SQ_INSTS_VALU"][1])
    out["waves_per_launch"] = round(split["SQ_WAVES"][1])
    return out


def per_sweep_block(cfg, sweep_ms, pmc):
    if not sweep_ms:
        return None
    w, h = cfg["w"], cfg["h"]
    algorithmic = 40.0 * w * h  # 8 planes read, 2 written
    phys = (pmc.get("hbm_read_bytes_per_launch") or 0) + (pmc.get("hbm_write_bytes_per_launch") or 0) or None
    alg_gbs = algorithmic / (sweep_ms * 1e-3) / 1e9
    phys_gbs = phys / (sweep_ms * 1e-3) / 1e9 if phys else None
    return {
        "kernel": pmc.get("kernel") or {0: "sweep_grey_kernel", 1: "sweep_grad_kernel", 3: "sweep_log_kernel"}.get(cfg["constancy"], "sweep kernel"),
        "bound": "hbm",
        "avg_launch_ms": round(sweep_ms, 5),
        "algorithmic_bytes_per_launch": algorithmic,
        # achieved / frac: the contract's ALGORITHMIC bytes (40 B per pixel, every plane once) over the launch
        # time -- what the sweep is worth, whatever the kernel re-reads
        "achieved": round(alg_gbs, 1),
        "achieved_is": "algorithmic bytes (40 B per pixel) over the launch time",
        "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(alg_gbs / HBM_PEAK_GBS, 4),
        # the counters' view of the same launches: FETCH_SIZE x correction + WRITE_SIZE.  Above the algorithmic bytes the
        # difference is halo rows and lanes that were re-read -- partly from the Infinity Cache, so physical_frac is a
        # fraction of memory-side traffic, not of HBM throughput
        "traffic": phys,
        "traffic_over_algorithmic": round(phys / algorithmic, 4) if phys else None,
        "physical_achieved": round(phys_gbs, 1) if phys_gbs else None,
        "physical_frac": round(phys_gbs / HBM_PEAK_GBS, 4) if phys_gbs else None,
        "mpix_sweeps_per_s": round(w * h / (sweep_ms * 1e-3) / 1e6, 1),
    }


def load_pmc(workload, algorithm):
    """Per-launch PMC figures of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/traffic.json, written by tools/pmc_traffic.py); {} if not measured for this workload."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return {}
    return table.get("%s/algorithm%d" % (workload, algorithm)) or {}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()[:16]


class Job:
    """One workload on this rank, run through the C++ batch entry (OpticalFlowBatch2D, host/optical_flow_batch_2d.cpp):
    lanes = independent (stream, OpticalFlow2D, plane pool) triples, every lane with its own copy of the planes it works
    on; consecutive steps rotate over the lanes, so one step's launch-bound coarse levels overlap another step's fine
    levels.  A step of a single-pair workload is that pair; a step of a batched workload is ONE lock-step group of all
    the rank's pairs (--batch-mode groups: the pairs one below the other in tall containers, every kernel launched
    once for the group) or, with --batch-mode lanes, the pairs spread one by one over the lanes."""

    def __init__(self, flow2d, batch, workload, cfg, args, rank, local_rank, world, out_tensor=None):
        self.flow2d, self.batch, self.workload, self.cfg, self.args = flow2d, batch, workload, cfg, args
        self.rank, self.world, self.local_rank = rank, world, local_rank
        w, h = cfg["w"], cfg["h"]
        self.single = cfg["pairs_per_rank"] == 1
        self.grouped = not self.single and args.batch_mode == "groups"
        self.group = cfg["pairs_per_rank"] if self.grouped else 1
        # Single-pair workloads up to 4096^2 (configs 2 and 3): consecutive STEPS are handed to the batch entry `step_group` at a time
        # as independent pairs, every one with a plane set of its own, and the C++ object forms a lock-step group of them itself
        # (OpticalFlowBatch2D::ComputeFlowBatchDeviceGrouped: one launch per kernel for the group; pairs that sit one container
        # apart -- how the sets are allocated below -- run in place, others are gathered and handed back).
        # A step is still one pair's whole pyramid; K steps are K pairs.
        self.step_group = 1
        if self.single:
            # (round 5, profiles/r05_experiments/step_group_sweeps.txt: 584 x 388 8 / 16 / 32 / 64 -> 1 707 / 1 973 / 2 340 / 2 007 pairs/s,
            #  1024^2 4 391 / 4 711 / 4 890 / 4 512, 4096^2 2 / 4 / 8 -> 332 / 338 / 345 at the driver's 20 steps and 340 / 345 / 348 at 100,
            #  8192^2 1 / 2 -> 96.4 / 94.3: the small kernels and coarse levels of a group share launches, and every launch of a
            #  graph costs its stream 4-5 us.  Round 6, groups in place -- no gather / hand-back copies: 8192^2 1 / 2 -> 98.3-98.6 / 99.2-99.3,
            #  4096^2 4 / 8 / 16 -> 352 / 356 / 357, 1024^2 16 / 32 / 64 -> 5 087 / 5 180 / 4 790)
            self.step_group = args.step_group if args.step_group > 0 else (32 if w * h <= 1024 * 1024 else 16 if w * h <= 2048 * 2048 else 8 if w * h <= 4096 * 4096 else 2 if w * h <= 8192 * 8192 else 1)
        self.pending = []
        self.rotate = self.single or self.grouped  # a step is one entry; steps rotate over the lanes
        self.n_lanes = max(1, min(args.max_lanes, args.pipeline if self.rotate else cfg["pairs_per_rank"] * args.pipeline))
        self.runner = flow2d.OpticalFlowBatch(w, h, cfg["constancy"], lanes=self.n_lanes, device=local_rank,
                                              group_size=self.group if self.step_group == 1 else self.step_group)
        self.ctx = flow2d.Context(local_rank)  # plane allocation, uploads, downloads (created after the lanes' streams)
        self.runner.use_graph(not args.no_graph)
        # rank 0's parameter block on every rank (RCCL broadcast), then the same solve everywhere
        block = batch.broadcast_params([cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"], 0.001,
                                        0.001, cfg["median"], cfg["sigma"], args.algorithm])
        self.params = self.runner.params(int(block[0]), block[1], int(block[2]), int(block[3]), block[4], block[5],
                                         block[6], int(block[7]), block[8], int(block[9]), sor_omega=cfg.get("sor_omega", 0.0))
        # this rank's pairs, resident in HBM before any timed region; pair k -> rank k mod world
        self.owned = batch.pairs_of_rank(cfg["pairs_per_rank"] * world, rank, world)
        frames = [workload_pair(workload, cfg, gk, world) for gk in self.owned]
        self.first_pair = frames[0]
        self.touched = set()  # plane sets written since the last poison()
        c, G = self.ctx, self.group
        # entries: (frame 0, frame 1, u, v, [global pair indices]); planes are G containers tall
        self.sets = []
        if self.step_group > 1:
            # per lane: step_group plane sets, every one of them a pair of its own (global pair rank + world * j for the
            # rank's set j; j = 0 is the workload's pair, which the CPU oracle checks) -- n_lanes x step_group DIFFERENT
            # pairs are in flight, not copies of one
            ids = [self.owned[0] + world * j for j in range(self.n_lanes * self.step_group)]
            more = workload_pairs(workload, cfg, ids[1:], world)
            pairs_all = [frames[0]] + more
            if args.scattered_groups:  # every plane an allocation of its own: the object gathers a group's frames and hands its flows back
                for j, (a, b) in enumerate(pairs_all):
                    self.sets.append((c.plane(w, h, a), c.plane(w, h, b), c.plane(w, h), c.plane(w, h), [ids[j]]))
            else:
                # a lane's step_group pairs one below the other in four allocations (frame 0, frame 1, u, v), every pair still a plane
                # set of its own to the batch entry: pairs that sit one container apart are a group as they lie, and the object
                # runs the pyramid on them in place (round 6: the gather / hand-back copies were 4-6 % of a group's time)
                S = self.step_group
                assert self.runner.group_stride == self.runner.pitch * h
                self.tall = []
                for lane in range(self.n_lanes):
                    mine = pairs_all[lane * S:(lane + 1) * S]
                    tall = (c.plane(w, h * S, np.vstack([q[0] for q in mine])), c.plane(w, h * S, np.vstack([q[1] for q in mine])),
                            c.plane(w, h * S), c.plane(w, h * S))
                    self.tall.append(tall)
                    for k in range(S):
                        at = k * self.runner.group_stride
                        self.sets.append(tuple(_Borrowed(c, t.ptr + at, t.pitch, w, h) for t in tall) + ([ids[lane * S + k]],))
            del more, pairs_all
        elif self.rotate:
            stacked = (np.vstack([f[0] for f in frames]), np.vstack([f[1] for f in frames]))
            for lane in range(self.n_lanes):
                if out_tensor is not None and lane == 0:  # lane 0 writes straight into the gather buffer [2, G, h, pitch]
                    assert out_tensor.shape[-1] * 4 == self.runner.pitch
                    pu = _Borrowed(c, out_tensor[0].data_ptr(), self.runner.pitch, w, h * G)
                    pv = _Borrowed(c, out_tensor[1].data_ptr(), self.runner.pitch, w, h * G)
                else:
                    pu, pv = c.plane(w, h * G), c.plane(w, h * G)
                self.sets.append((c.plane(w, h * G, stacked[0]), c.plane(w, h * G, stacked[1]), pu, pv, list(self.owned)))
        else:
            for n, gk in enumerate(self.owned):
                if out_tensor is not None:
                    assert out_tensor.shape[-1] * 4 == self.runner.pitch
                    pu = _Borrowed(c, out_tensor[0, n].data_ptr(), self.runner.pitch, w, h)
                    pv = _Borrowed(c, out_tensor[1, n].data_ptr(), self.runner.pitch, w, h)
                else:
                    pu, pv = c.plane(w, h), c.plane(w, h)
                self.sets.append((c.plane(w, h, frames[n][0]), c.plane(w, h, frames[n][1]), pu, pv, [gk]))
        c.synchronize()

    def sync(self):
        self.runner.synchronize()
        self.ctx.synchronize()

    def _queue(self, sets, first_lane):
        cols = [[q[i].ptr for q in sets] for i in range(4)]
        self.runner.compute_flow_batch_device(*cols, self.params, first_lane=first_lane)

    def flush(self):
        """Hands the collected steps of a step group to the batch entry (a partial group at the end of a region)."""
        if not self.pending:
            return
        n, lane = len(self.pending), (self.pending[0] // self.step_group) % self.n_lanes
        sets = self.sets[lane * self.step_group:lane * self.step_group + n]
        cols = [[q[i].ptr for q in sets] for i in range(4)]
        self.runner.compute_flow_batch_device_grouped(*cols, self.params, first_lane=lane)
        self.touched.update(range(lane * self.step_group, lane * self.step_group + n))
        self.pending = []

    def step(self, index):
        """One pass over this rank's pairs: ONE call into the C++ batch entry (replayed HIP graphs unless --no-graph)."""
        if self.step_group > 1:
            self.pending.append(index)
            if len(self.pending) == self.step_group:
                self.flush()
        elif self.rotate:
            lane = index % self.n_lanes
            self._queue([self.sets[lane]], lane)
            self.touched.add(lane)
        else:
            self._queue(self.sets, 0)
            self.touched.update(range(len(self.sets)))

    def poison(self):
        """Every flow plane of the job filled with 0x7f bytes (3.39e38): what a step does not write stays recognisable."""
        self.sync()
        for _, _, pu, pv, _ in self.sets:
            for q in (pu, pv):
                self.flow2d.Plane.fill_bytes(q, 0x7f)
        self.sync()
        self.touched = set()

    def eager_pass(self):
        """Every plane set once, launched eagerly (no graph): the recomputation the output check compares with."""
        self.runner.use_graph(False)
        if self.step_group > 1:
            for k in range(self.n_lanes * self.step_group):
                self.step(k)
            self.flush()
        else:
            self._queue(self.sets, 0)
        self.sync()
        self.runner.use_graph(not self.args.no_graph)

    def digests(self):
        """{(set index, pair index): (sha(u), sha(v))} of what is in HBM now."""
        self.sync()
        h = self.cfg["h"]
        out = {}
        for si, (_, _, pu, pv, gks) in enumerate(self.sets):
            u, v = pu.download(), pv.download()
            for k, gk in enumerate(gks):
                out[(si, gk)] = (sha(u[k * h:(k + 1) * h]), sha(v[k * h:(k + 1) * h]))
        return out

    def close(self):
        self.runner.close()
        self.ctx.close()


class _Borrowed:
    """A plane living in someone else's allocation (a torch tensor): same download interface as flow2d.Plane."""

    def __init__(self, ctx, ptr, pitch, width, height):
        self.ctx, self.ptr, self.pitch, self.width, self.height = ctx, ptr, pitch, width, height

    def download(self):
        return importlib.import_module("cuda-flow2d_amd").Plane.download(self)

    def fill_bytes(self, value):
        return importlib.import_module("cuda-flow2d_amd").Plane.fill_bytes(self, value)


def timed_region(job, batch, torch, steps, warmup, repeats=1):
    """Warm-up, poison, then `repeats` timed regions of EXACTLY `steps` steps each (barrier + device synchronise on both
    sides, max over ranks); returns the list of their wall times."""
    def barrier():
        batch.barrier()
        job.sync()
        torch.cuda.synchronize()

    for k in range(max(warmup, 1) * (job.n_lanes if job.rotate else 1) * job.step_group):
        job.step(k)  # also records the graphs of every lane
    job.flush()
    # K steps that are no multiple of the lock-step group end in a smaller group: its graph (another plane list, another
    # tile / strip mix) is recorded here, not inside the timed region
    whole = steps // job.step_group * job.step_group
    for k in range(whole, steps):
        job.step(k)
    job.flush()
    barrier()
    # the flow planes are poisoned between warm-up and the timed region: the digests the output check takes afterwards
    # are of values the K timed steps wrote (a replay that launched nothing would leave 0x7f7f7f7f behind)
    job.poison()
    times = []
    for _ in range(max(1, repeats)):
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            job.step(k)
        job.flush()
        barrier()
        times.append(batch.max_over_ranks(time.perf_counter() - t0))
    return times


def output_check(job):
    """The timed steps' results against an eager recomputation; single-pair workloads: all streams agree."""
    touched = set(job.touched)  # the plane sets the timed steps wrote (the others still hold the poison)
    replayed = {k: d for k, d in job.digests().items() if k[0] in touched}
    poison = sha(np.frombuffer(b"\x7f" * (job.cfg["w"] * job.cfg["h"] * 4), np.float32))
    poisoned = [k for k, d in replayed.items() if poison in d]
    job.eager_pass()
    eager = {k: d for k, d in job.digests().items() if k[0] in touched}
    lanes_identical = True
    if job.rotate:  # every lane holds a copy of the same pairs: pair by pair the lanes must agree
        by_pair = {}
        for (si, gk), d in replayed.items():
            by_pair.setdefault(gk, set()).add(d)
        lanes_identical = all(len(ds) == 1 for ds in by_pair.values())
    return {
        "ok": bool(replayed and replayed == eager and lanes_identical and not poisoned),
        "graph_replay_equals_eager": bool(replayed == eager),
        "planes_poisoned_before_timed_region": True,
        "plane_sets_written_by_timed_steps": len(touched),
        "streams_identical": bool(lanes_identical) if job.rotate and job.step_group == 1 else None,
        "distinct_pairs_in_flight": len({gk for (_, gk) in replayed}),
        "fields_hashed": 2 * len(replayed),
        "sha256_u_v_first_pair": list(replayed[min(replayed)]) if replayed else None,
    }


def pcie_ceiling(flow2d, local_rank, w=4096, h=4096, copies=16):
    """The box's own page-locked copy rates through the C-ABI, GB/s per direction: upload alone, download alone, both at once on
    two streams (what the host-entry leg's pcie_gbs_each_way is a fraction of).  None when page-locked memory is refused."""
    L = flow2d.hip_lib()
    up, down = flow2d.Context(local_rank), flow2d.Context(local_rank)
    imgs = []
    try:
        imgs = [flow2d.HostImage(w, h, True) for _ in range(4)]
        planes = [up.plane(w, h) for _ in range(4)]
        nbytes = w * h * 4

        def h2d(n):
            for k in range(n):
                L.flow2d_copy_h2d_2d(up.handle, planes[k % 2].ptr, planes[0].pitch, imgs[k % 2].array.ctypes.data, w * 4, w * 4, h)

        def d2h(n):
            for k in range(n):
                L.flow2d_copy_d2h_2d(down.handle, imgs[2 + k % 2].array.ctypes.data, w * 4, planes[2 + k % 2].ptr, planes[0].pitch, w * 4, h)

        out = {}
        for name, fn in (("upload_alone", lambda: h2d(copies)), ("download_alone", lambda: d2h(copies)),
                         ("both_at_once_each_way", lambda: (h2d(copies), d2h(copies)))):
            fn()
            up.synchronize(), down.synchronize()
            t0 = time.perf_counter()
            fn()
            up.synchronize(), down.synchronize()
            out[name] = round(copies * nbytes / (time.perf_counter() - t0) / 1e9, 1)
        return out
    except Exception as e:  # noqa: BLE001 - an informational leg
        print("bench: pcie ceiling: %r" % (e,), file=sys.stderr)
        return None
    finally:
        for q in imgs:
            q.close()
        up.close(), down.close()


def host_entry_leg(job, batch, torch, steps):
    """SURVEY 8(d) metric 2 in the reference's own bracket (optical_flow_2d.cpp:173-179,214-215,544-554): host images in,
    host flows out, uploads and downloads INSIDE the timed region.  OpticalFlowBatch2D::ComputeFlowBatch queues upload,
    pyramid and download of an entry on its lane's stream; the lanes overlap, so one lane's DMA runs beside the others' kernels.  The
    images are Data2D objects in page-locked memory (HostMemory::Pinned, the reference's ALLOCATE_PINNED_MEMORY option).
    Every step takes the rank's pairs from the same host frames and delivers into its own flow images (2 x lanes sets,
    reused round-robin).  Timed like the main region: barrier + synchronise on both sides, max over ranks."""
    flow2d, cfg = job.flow2d, job.cfg
    w, h, G = cfg["w"], cfg["h"], cfg["pairs_per_rank"]
    frames = [workload_pair(job.workload, cfg, gk, job.world) for gk in job.owned]
    f0s = [flow2d.HostImage(w, h, True, f[0]) for f in frames]
    f1s = [flow2d.HostImage(w, h, True, f[1]) for f in frames]
    # steps handed over together (the batch object forms a lock-step group of them).  With host copies inside the bracket a large
    # group delays its first kernel behind all of its uploads and its downloads behind its last kernel: smaller groups than the
    # device-resident region's above 512^2 (4096^2, groups of 8 / 2: 273 / 322 pairs/s; 1024^2, 32 / 8: 2 966 / 3 724; 584 x 388
    # is launch-bound either way: 32 / 8: 2 122 / 1 769)
    N = job.step_group if w * h <= 512 * 512 else min(job.step_group, 8 if w * h <= 1024 * 1024 else 4 if w * h <= 2048 * 2048 else 2)
    n_sets = 2 * max(job.n_lanes, job.args.host_entry_lanes or 0, 6) * N
    steps = max(N, steps // N * N)
    outs = [([flow2d.HostImage(w, h, True) for _ in range(G)], [flow2d.HostImage(w, h, True) for _ in range(G)])
            for _ in range(n_sets)]
    images = f0s + f1s + [q for us, vs in outs for q in us + vs]
    pinned = all(q.pinned for q in images)
    # (the host-image entry takes whole lock-step groups only: a smaller group needs a batch object of its own)
    # (with copies inside the bracket a lane spends a third of its cycle transferring: six lanes keep four computing -- 4096^2,
    #  4 / 6 / 8 lanes: 305 / 318-319 / 297-299 pairs/s, round 6; small frames are launch-bound and keep the region's lanes)
    host_lanes = job.args.host_entry_lanes or (6 if w * h >= 2048 * 2048 and job.n_lanes == 4 else job.n_lanes)
    own_runner = (job.step_group > 1 and N != job.step_group) or host_lanes != job.n_lanes
    runner = flow2d.OpticalFlowBatch(w, h, cfg["constancy"], lanes=host_lanes, device=job.local_rank, group_size=N) if own_runner else job.runner
    runner.use_graph(not job.args.no_graph)

    def call(c):
        """steps c*N .. c*N + N-1 in one call of the batch entry"""
        us = [q for j in range(N) for q in outs[(c * N + j) % n_sets][0]]
        vs = [q for j in range(N) for q in outs[(c * N + j) % n_sets][1]]
        # one entry (a pair, or a lock-step group) per call, calls rotate over the lanes; lanes mode: the pairs spread from lane 0
        runner.compute_flow_batch(f0s * N, f1s * N, us, vs, job.params, first_lane=(c % host_lanes) if job.rotate else 0)

    def barrier():
        batch.barrier()
        runner.synchronize()
        torch.cuda.synchronize()

    try:
        for c in range(n_sets // N):  # warm-up: allocates the staging planes, records the graphs of both slots of every lane
            call(c)
        barrier()
        t0 = time.perf_counter()
        for c in range(steps // N):
            call(c)
        barrier()
        elapsed = batch.max_over_ranks(time.perf_counter() - t0)
        # every delivered flow image against the device-resident result of the same pair (sha of the timed region's fields)
        want = {gk: d for (si, gk), d in sorted(job.digests().items(), reverse=True)}  # (a pair on several sets: the first set's)
        same = all((sha(us[i].array), sha(vs[i].array)) == want[gk]
                   for us, vs in outs[:min(n_sets, steps)] for i, gk in enumerate(job.owned))
        pairs = steps * G * job.world
        bytes_per_pair = 4 * w * h * 4
        return {
            "pairs_per_s": round(pairs / elapsed, 3), "ms_per_step": round(elapsed / steps * 1e3, 4), "steps": steps,
            "bracket": "host Data2D frames in -> host Data2D flows out (upload, pyramid, download), as the reference's "
                       "own timer brackets ComputeFlow",
            "host_path": "OpticalFlowBatch2D::ComputeFlowBatch (C++): upload, pyramid (graph replay), download on each of "
                         "%d lanes' own stream; the lanes overlap" % host_lanes,
            "host_memory": "page-locked Data2D" if pinned else "pageable Data2D (pinned allocation failed)",
            "pcie_bytes_per_pair": bytes_per_pair,
            "pcie_gbs_each_way": round(pairs / job.world * bytes_per_pair / 2 / elapsed / 1e9, 2),
            # ... against what this box's page-locked copies reach with nothing else running (tools/pcie_pinned.py's measurement)
            "pcie_ceiling_gbs": pcie_ceiling(job.flow2d, job.local_rank) if job.rank == 0 else None,
            "flows_bit_identical_to_device_resident_run": bool(same),
        }
    finally:
        for q in images:
            q.close()
        if own_runner:
            runner.close()


def oracle_check(job):
    """The line's own parity bit: the first pair's flow as the timed region left it in HBM against the CPU oracle
    (oracle/flow2d_oracle.c, the checker -- never the thing measured) on the same pair, every pixel, bit for bit.
    Rank 0, after the timed region; a few seconds of OpenMP at 4096^2."""
    from oracle import oracle as O

    cfg = job.cfg
    O.lib()
    h = cfg["h"]
    _, _, pu, pv, _ = job.sets[0]
    u, v = pu.download()[:h], pv.download()[:h]
    t0 = time.perf_counter()
    ou, ov, t_finest = O.compute_flow(job.first_pair[0], job.first_pair[1], cfg["levels"], cfg["scale"], cfg["outer"],
                                      cfg["inner"], cfg["alpha"], 0.001, 0.001, cfg["median"], cfg["sigma"], cfg["constancy"],
                                      sor_omega=cfg.get("sor_omega", 0.0))
    seconds = time.perf_counter() - t0
    return {"first_pair_equals_cpu_oracle": bool(np.array_equal(u, ou) and np.array_equal(v, ov)),
            "pixels_compared": int(2 * u.size), "oracle_seconds": round(seconds, 2),
            "max_abs_difference": float(max(np.abs(u - ou).max(), np.abs(v - ov).max())),
            "_timing": (seconds, t_finest)}


def probe_builds(args, cfg):
    """Level-solve time / 10 of the strip kernel's timing probes (developer libraries ab/memory.so, ab/compute.so: wrong results by
    design, never loaded into this process) at the workload's finest level, each in a child process of its own."""
    out = {}
    root = os.path.dirname(os.path.abspath(__file__))
    for key, lib in (("memory_only_us", "memory"), ("compute_only_us", "compute")):
        path = os.path.join(root, "ab", lib + ".so")
        if not os.path.exists(path):
            continue
        try:
            r = subprocess.run([sys.executable, os.path.join(root, "tools", "time_sweep.py"), str(cfg["w"]), str(cfg["h"]), "2", str(min(cfg["inner"], 5))],
                               env=dict(os.environ, FLOW2D_HIP_LIB=path), capture_output=True, text=True, timeout=120)
            lines = [l for l in r.stdout.splitlines() if l.startswith("constancy %d " % (1 if cfg["constancy"] == 1 else 0))]
            out[key] = round(float(lines[-1].split("level solve")[1].split("ms")[0]) * 100.0, 1)  # ms per 10 launches -> us per launch
        except Exception as e:  # noqa: BLE001 - an informational leg
            print("bench: timing probe %s: %r" % (lib, e), file=sys.stderr)
    return out


def roofline_sample(job, passes=3):
    """Eager passes of the first pair on a stream of its own, alone on the GPU, with HIP events on that stream around
    every level's solve and every finest-level solver launch (flow2d_timing_enable mode 2)."""
    flow2d, cfg = job.flow2d, job.cfg
    w, h = cfg["w"], cfg["h"]
    job.sync()
    c = flow2d.Context(job.local_rank)
    flow = flow2d.OpticalFlow(w, h, cfg["constancy"], ctx=c)
    try:
        f0, f1 = (c.plane(w, h, a) for a in job.first_pair)
        u, v = c.plane(w, h), c.plane(w, h)
        # the CPU-side checks before this leg leave the GPU idle for seconds: a few untimed passes bring its clocks back
        # up before the sampled ones
        for _ in range(4):
            flow.compute_flow_device(f0.ptr, f1.ptr, u.ptr, v.ptr, job.params, 0)
        c.synchronize()
        flow.reset_timings()
        for _ in range(passes):
            flow.compute_flow_device(f0.ptr, f1.ptr, u.ptr, v.ptr, job.params, 2)
        c.synchronize()
        # The clock the chip HOLDS under the finest level's solver launches: three level solves of the workload's counts back to
        # back on this stream, bracketed by one sleeping wave per XCD on a context of its own (flow2d_clock_probe_start).
        probe = flow2d.Context(job.local_rank)
        try:
            scratch = [c.plane(w, h).fill_bytes(0) for _ in range(6)]

            def level_solve():
                c.solve_level(f0, f1, u, v, *scratch, w, h, 1.0, 1.0, cfg["alpha"], 0.001, 0.001, cfg["outer"], cfg["inner"],
                              cfg["constancy"], flow2d.SOLVER_AUTO, sor_omega=cfg.get("sor_omega", 0.0))

            u.fill_bytes(0), v.fill_bytes(0)
            level_solve()
            e0, e1 = c.event(), c.event()
            c.record(e0)
            level_solve()
            c.record(e1)
            solve_ms = c.elapsed_ms(e0, e1)
            probe.clock_probe_start(max(100.0, 0.85 * 3 * solve_ms * 1e3))
            for _ in range(3):
                level_solve()
            c.synchronize()
            job.sample_clock_ghz = [round(g, 4) for g in probe.clock_probe_read()]
        except Exception as e:  # noqa: BLE001 - an informational leg
            print("bench: clock probe: %r" % (e,), file=sys.stderr)
        finally:
            probe.close()
        finest = [r for r in flow.level_timings() if (r[0], r[1]) == (w, h)]
        # one pair alone on the GPU, launch to done, replayed from its graph (no timing events inside): the latency
        # a single pair sees, as opposed to the pipelined rate of the timed region
        def replayed_ms(obj, cx=None):
            cx = cx or c
            e0, e1 = cx.event(), cx.event()
            cx.record(e0)
            obj.compute_flow_device(f0.ptr, f1.ptr, u.ptr, v.ptr, job.params, 0)
            cx.record(e1)
            cx.synchronize()
            return cx.elapsed_ms(e0, e1)

        # ... and the same pair through an object that behaves like a lane of a pipeline (OpticalFlow2D::lone = false: the
        # pipeline's build of the strip kernel everywhere): what the packed build buys a lone pair.  The two take turns (the chip's clock ramps over tens of milliseconds: whoever is measured later would look better) and
        # each reports the median of its replays after the recording one.
        # (`lone` is a property of the CONTEXT -- flow2d_context_set_lone -- so each object gets a context of its own)
        c_plain = flow2d.Context(job.local_rank)
        plain = flow2d.OpticalFlow(w, h, cfg["constancy"], ctx=c_plain, lone=False)
        try:
            objs = ((flow, c), (plain, c_plain))
            for o, cx in objs:
                o.use_graph(True)
                replayed_ms(o, cx)  # these record
            times = [[replayed_ms(o, cx) for o, cx in objs] for _ in range(9)]
            latency, job.single_stream_latency_ms = (float(np.median([t[k] for t in times])) for k in range(2))
        finally:
            plain.close()
            c_plain.close()
        return finest, latency
    finally:
        flow.close()
        c.close()


def per_sweep_sample(job, launches=10, rounds=4):
    """SURVEY 8(d) metric 1, second half: the pure inner-sweep rate.  The per-sweep Jacobi kernel of the workload's data
    term (sweep_grey / sweep_grad / sweep_log: one launch per reference launch of solve_2d*, cuda_operation_solve_2d.cpp:263-289)
    at the workload's finest level, alone on the GPU: HIP events on the launch stream around `launches` back-to-back
    launches, mean of the rounds after the first.  Returns ms per launch."""
    flow2d, cfg = job.flow2d, job.cfg
    w, h = cfg["w"], cfg["h"]
    job.sync()
    c = flow2d.Context(job.local_rank)
    try:
        f0, f1 = (c.plane(w, h, a) for a in job.first_pair)
        u, v, du, dv, phi, ksi, tdu, tdv = (c.plane(w, h).fill_bytes(0) for _ in range(8))
        c.compute_phi_ksi(f0, f1, u, v, du, dv, w, h, 1.0, 1.0, 0.001, 0.001, phi, ksi)
        ms = []
        for _ in range(rounds):
            e0, e1 = c.event(), c.event()
            c.record(e0)
            for k in range(launches):  # ping-pong like CudaOperationSolve2D::Execute does between du/dv and the temporaries
                a, b = ((du, dv), (tdu, tdv)) if k % 2 == 0 else ((tdu, tdv), (du, dv))
                c.solve_sweep(f0, f1, u, v, a[0], a[1], phi, ksi, w, h, 1.0, 1.0, cfg["alpha"], b[0], b[1], cfg["constancy"])
            c.record(e1)
            ms.append(c.elapsed_ms(e0, e1) / launches)
        return float(np.mean(ms[1:]))
    except Exception:  # noqa: BLE001 - an informational leg
        return None
    finally:
        c.close()


def sor_time_to_residual(flow2d, cfg, local_rank, pair, omegas=(1.0, 1.5, 1.9), max_iterations=6):
    """Jacobi against red-black SOR on the workload's WHOLE pyramid (one pair alone on the GPU): the reference's counts
    (outer x 5 Jacobi sweeps per level) against SOR with omega in `omegas` and 1, 2, ... iterations per outer iteration, the
    same outer iterations, levels, median and blur.  Yardsticks for a flow (u, v):
      error_to_converged -- rms distance to the flow of the same pyramid solved (nearly) to convergence on every level: 40 outer
                            x 12 SOR(1.5) iterations; a 60 x 16 run differs from it by `reference_floor`.  "Reaching Jacobi"
                            = an error no larger than that of the Jacobi run.
      endpoint_error     -- mean |(u, v) - (dx, dy)| against the synthetic pair's true motion, for scale: it holds
                            the model's error as well as the solver's.
    ms: HIP events around one replayed pyramid, best of three after the recording pass."""
    w, h = cfg["w"], cfg["h"]
    c = flow2d.Context(local_rank)
    flow = flow2d.OpticalFlow(w, h, cfg["constancy"], ctx=c)
    try:
        f0, f1 = (c.plane(w, h, a) for a in pair)
        u, v = c.plane(w, h), c.plane(w, h)
        flow.use_graph(True)

        def run(outer, inner, omega, reps=4):
            p = flow.params(cfg["levels"], cfg["scale"], outer, inner, cfg["alpha"], 0.001, 0.001, cfg["median"], cfg["sigma"],
                            0, sor_omega=omega)
            best = None
            for rep in range(reps):  # the first call records the graph
                e0, e1 = c.event(), c.event()
                c.record(e0)
                flow.compute_flow_device(f0.ptr, f1.ptr, u.ptr, v.ptr, p, 0)
                c.record(e1)
                ms = c.elapsed_ms(e0, e1)
                best = ms if (rep or reps == 1) and (best is None or ms < best) else best
            return (u.download(w, h).astype(np.float64), v.download(w, h).astype(np.float64)), best

        def rms(a, b):
            return float(np.sqrt(np.mean(np.concatenate([(a[0] - b[0]).ravel(), (a[1] - b[1]).ravel()]) ** 2)))

        def aee(a):
            return float(np.mean(np.hypot(a[0] - cfg["dx"], a[1] - cfg["dy"])))

        converged, _ = run(40, 12, 1.5, reps=1)
        floor = rms(run(60, 16, 1.5, reps=1)[0], converged)
        jac, ms_j = run(cfg["outer"], 5, 0.0)
        err_j = rms(jac, converged)
        out = {"pyramid": "%dx%d, %d levels, %d outer iterations per level, %s data term, alpha %g, median %d" %
                          (w, h, cfg["levels"], cfg["outer"], CONSTANCY_NAME[cfg["constancy"]], cfg["alpha"], cfg["median"]),
               "error_to_converged": "rms distance of (u, v) to the same pyramid at 40 outer x 12 SOR(1.5) iterations per level",
               "reference_floor": floor, "converged_endpoint_error": aee(converged),
               "jacobi": {"sweeps_per_outer": 5, "ms_per_pair": round(ms_j, 4), "error_to_converged": err_j,
                          "endpoint_error": aee(jac)}, "sor": []}
        for omega in omegas:
            rows, reached = [], None
            for n in range(1, max_iterations + 1):
                f, ms = run(cfg["outer"], n, omega)
                rows.append({"iterations_per_outer": n, "ms_per_pair": round(ms, 4), "error_to_converged": rms(f, converged),
                             "endpoint_error": aee(f)})
                if rows[-1]["error_to_converged"] <= err_j:
                    reached = n
                    break
            out["sor"].append({"omega": omega, "iterations_per_outer_to_reach_jacobi": reached,
                               "ms_per_pair_to_reach_jacobi": rows[-1]["ms_per_pair"] if reached else None,
                               "speedup_over_jacobi": round(ms_j / rows[-1]["ms_per_pair"], 3) if reached else None, "runs": rows})
        return out
    finally:
        flow.close()
        c.close()


def batch_leg(flow2d, batch, torch, args, rank, local_rank, world):
    """BASELINE.json configs[3]: 8 pairs of 1920x1080 per GPU (64 on 8 GPUs), then ONE all_gather of the flow fields."""
    cfg = WORKLOADS[BATCH_WORKLOAD]
    w, h = cfg["w"], cfg["h"]
    pitch_floats = flow2d.hip_lib().flow2d_plane_pitch_bytes(w) // 4
    n_local = cfg["pairs_per_rank"]
    # u planes of the rank's pairs, then their v planes: [2, pairs, H, pitch] (a lock-step group's tall containers)
    local = torch.zeros((2, n_local, h, pitch_floats), dtype=torch.float32, device=torch.device("cuda", local_rank))
    job = Job(flow2d, batch, BATCH_WORKLOAD, cfg, args, rank, local_rank, world, out_tensor=local)
    # at least 64 steps (like the host-entry leg): over 20 steps the fill and drain of the four lanes weigh 12 % (round 4:
    # 2 092 pairs/s in the driver's line against 2 388 for the same workload at 100 steps)
    steps = max(64, args.steps)
    # (median of three regions after the same warm-up as the main region's: one region right after the CPU-side legs read low)
    elapsed = float(np.median(timed_region(job, batch, torch, steps, args.warmup, 3)))
    check = output_check(job)
    # gather: every rank's [2, 8, H, pitch] block -> [world, 2, 8, H, pitch]; pair k = [k % world, :, k // world].  Two forms,
    # each run once untimed and once timed: to every rank (all_gather) and to rank 0 only (gather: what BASELINE.json asks)
    def timed(collective, out):
        torch.cuda.synchronize()
        batch.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = collective(local, out=out)
        torch.cuda.synchronize()
        return out, batch.max_over_ranks(time.perf_counter() - t0)

    gathered, _ = timed(batch.all_gather_fields, None)
    gathered, all_gather_s = timed(batch.all_gather_fields, gathered)
    at_root, _ = timed(batch.gather_fields_to_root, None)
    at_root, gather_s = timed(batch.gather_fields_to_root, at_root)
    mine_ok = bool(torch.equal(gathered[rank], local))
    nonzero = bool((gathered.abs().amax(dim=(3, 4)) > 0).all().item())
    root_ok = bool(torch.equal(at_root, gathered)) if rank == 0 else at_root is None
    job.close()
    pairs = steps * n_local * world
    px_iters = float(w) * h * cfg["outer"] * cfg["inner"]
    return {
        "workload": BATCH_WORKLOAD, "pairs_per_gpu": n_local, "pairs_total_per_step": n_local * world, "steps": steps,
        "value": round(px_iters * pairs / elapsed / 1e6, 1), "unit": "Mpixel*iters/s",
        "pairs_per_s": round(pairs / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 3),
        "streams_per_gpu": job.n_lanes, "batch_mode": "lock-step group of %d pairs per launch" % job.group if job.grouped
        else "pairs spread over the lanes",
        "gather": {"collective": "gather to rank 0 (RCCL grouped send/recv)" if world > 1 else "none (one process)",
                   "bytes_per_rank": int(local.numel() * 4), "ms": round(gather_s * 1e3, 3),
                   "all_gather_into_tensor_ms": round(all_gather_s * 1e3, 3),
                   "own_block_intact": mine_ok, "every_pair_present": nonzero, "root_equals_all_gather": root_ok},
        "output_check": check,
    }


def spawn_ranks(args):
    """`python bench.py --gpus N` from a plain shell (no WORLD_SIZE in the environment): start the N ranks as FRESH child
    processes -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` -- before
    this process has made any HIP / torch.cuda call, let rank 0's JSON line through (the children inherit stdout) and
    exit with the launcher's status.  Nothing is exec'd and nothing is re-launched after GPU initialisation."""
    import socket
    import subprocess

    if not (args.plumbing_check or args.rehearse_on_one_gpu):
        import torch  # device_count() alone does not initialise the GPU on this image
        visible = torch.cuda.device_count()
        if visible < args.gpus:
            sys.exit("bench.py --gpus %d: %d HIP device(s) visible; the flow2d path has no CPU fallback" %
                     (args.gpus, visible))
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def plumbing_check(batch, args):
    """--plumbing-check: the rank plumbing of a multi-GPU run WITHOUT the flow computation, on the gloo backend and host
    tensors, so that it runs on a box without GPUs (tests/test_batch_gloo.py runs `bench.py --gpus 2 --plumbing-check`
    from a plain shell): launcher -> process group -> parameter broadcast -> barriers around a timed region -> max
    over ranks -> all_gather and gather-to-root of a stand-in field block.  Prints a line that carries no metric: it
    says that the ranks started and talked, nothing about the product."""
    import torch

    rank, _, world = batch.init(backend="gloo")
    cfg = WORKLOADS[args.workload]
    block = batch.broadcast_params([cfg["levels"], cfg["scale"], cfg["outer"], cfg["inner"], cfg["alpha"]] if rank == 0
                                   else [0, 0, 0, 0, 0])
    owned = batch.pairs_of_rank(cfg["pairs_per_rank"] * world, rank, world)
    batch.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    batch.barrier()
    slowest = batch.max_over_ranks(time.perf_counter() - t0)
    local = torch.full((2, len(owned), 4, 8), float(rank + 1))  # the shape of a rank's flow block: [2, pairs, H, pitch]
    everywhere = batch.all_gather_fields(local)
    at_root = batch.gather_fields_to_root(local)
    expect = torch.arange(1, world + 1, dtype=torch.float32).view(world, 1, 1, 1, 1).expand(world, *local.shape)
    ok = bool(torch.equal(everywhere, expect)) and (rank != 0 or bool(torch.equal(at_root, expect))) and \
        (rank == 0 or at_root is None)
    oks = batch.gather_digests({rank: float(ok)}, world)
    if rank == 0:
        print(json.dumps({"plumbing_check": True, "n_gpus": world, "backend": "gloo", "params": block,
                          "pairs_of_rank_0": owned, "slowest_rank_s": round(slowest, 4),
                          "collectives_ok": bool(all(v == 1.0 for v in oks))}))
    batch.shutdown()
    if not ok:
        sys.exit("bench.py --plumbing-check: a collective delivered the wrong data on rank %d" % rank)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of K steps each after the one warm-up; the line reports their median")
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--algorithm", type=int, default=0, help="flow2d_solver_algorithm: 0 auto, 1 per-sweep, 2 fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-baseline", action="store_true")
    ap.add_argument("--no-batch-leg", action="store_true")
    ap.add_argument("--host-entry-lanes", type=int, default=0, help=argparse.SUPPRESS)  # (developer A/B: lanes of the host-entry leg's own runner)
    ap.add_argument("--no-probe-builds", action="store_true", help="skip the memory-only / compute-only timing probes (child processes on ab/*.so)")
    ap.add_argument("--batch-leg-last", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-oracle-check", action="store_true",
                    help="skip comparing the first pair's flow with the CPU oracle (output_check.oracle)")
    ap.add_argument("--no-host-entry-leg", action="store_true", help="skip the H<->D-inclusive leg (pairs_per_s_incl_h2d)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying HIP graphs")
    ap.add_argument("--pipeline", type=int, default=4,
                    help="independent pairs in flight per GPU when a step holds a single pair: consecutive steps go to "
                         "alternating streams so one pair's launch-bound coarse levels overlap the next pair's fine levels")
    ap.add_argument("--max-lanes", type=int, default=4, help="upper bound on the lanes (streams) per GPU")
    ap.add_argument("--scattered-groups", action="store_true",
                    help="single-pair workloads with step groups: every plane of every pair an allocation of its own (the object gathers a "
                         "group's frames into staging containers and hands the flows back: two device copies per group) instead of a lane's "
                         "pairs one below the other in four allocations, which the object runs in place")
    ap.add_argument("--step-group", type=int, default=0,
                    help="single-pair workloads: consecutive steps handed to the batch entry this many at a time, which "
                         "forms a lock-step group of them (0 = automatic: 32 up to 1024^2, 16 up to 2048^2, 8 up to 4096^2, 2 up to 8192^2, else 1; the "
                         "finest level of a 4096^2 group still runs one launch per pair)")
    ap.add_argument("--batch-mode", choices=["groups", "lanes"], default="groups",
                    help="batched workloads: all pairs of a step as one lock-step group (every kernel launched once for "
                         "the group) or spread one by one over the lanes")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 --pmc child passes that measure the dominant kernel's HBM bytes and VALU "
                         "instructions for this run (N = 1 only); the line then quotes profiles/traffic.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="developer: the N ranks of --gpus N all use GPU 0 and talk over gloo (keep N <= 6), to walk the "
                         "multi-rank branches of this file on a one-GPU box; the line says so and is NOT a measurement")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="rank plumbing only (launcher, process group on gloo, broadcast, barriers, gathers) without the "
                         "flow computation; needs no GPU and prints no metric")
    args = ap.parse_args()
    cfg = WORKLOADS[args.workload]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)  # does not return

    batch = importlib.import_module("cuda-flow2d_amd.batch")
    if args.plumbing_check:
        return plumbing_check(batch, args)
    flow2d = importlib.import_module("cuda-flow2d_amd")
    if args.pmc_child:
        return pmc_child(flow2d, args, cfg)
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
    args.gpus = world  # launched by torch.distributed.run: the launcher's world size is the number of GPUs
    if args.rehearse_on_one_gpu:
        if world > 6:
            sys.exit("bench.py --rehearse-on-one-gpu: at most 6 ranks may share a GPU")
        local_rank = 0
    # counters of THIS workload, collected by child processes before this one initialises the GPU
    pmc_run = pmc_passes(args, cfg) if (world == 1 and not args.no_pmc) else {}

    import torch  # device plumbing only: barrier, device-wide synchronise, max-over-ranks, the RCCL gather buffer

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the flow2d path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # RCCL; no-op for one process (the rehearsal's ranks share a GPU, which RCCL refuses: gloo)
    batch.init(backend="gloo" if args.rehearse_on_one_gpu else "nccl", device=torch.device("cuda", local_rank))

    w, h = cfg["w"], cfg["h"]
    # The batch leg (BASELINE.json configs[3]) runs FIRST, on a GPU nothing else has touched yet, and is gone before the main job
    # allocates: behind the main job's legs it read low by amounts that changed with what ran before it (after the sampling legs
    # 8-9 % in rounds 4 and 5; behind a main job of eight-pair groups and its host-entry leg 2 281 against 2 530-2 630 pairs/s:
    # profiles/r05_experiments/batch_leg_bisect.txt, batch_leg_first_ab.txt).  The line's `value` is the main job's region below.
    batch_result = None
    if not args.no_batch_leg and args.workload != BATCH_WORKLOAD and not args.batch_leg_last:
        batch_result = batch_leg(flow2d, batch, torch, args, rank, local_rank, world)
    job = Job(flow2d, batch, args.workload, cfg, args, rank, local_rank, world)
    free_b, total_b = job.ctx.mem_info()
    plane_b = job.runner.pitch * h * (job.group if job.step_group == 1 else job.step_group)  # a (group-tall) container
    memory_parts = {
        "container_mib": round(job.runner.pitch * h / 2 ** 20, 1),
        "plane_pools_gib": round(job.n_lanes * 14 * plane_b / 2 ** 30, 3),  # per lane: the 12 containers + 2 packed planes
        "caller_planes_gib": round(sum(q.pitch * q.height for q in {q for s in job.sets for q in s[:4]}) / 2 ** 30, 3),
        "what": "%d lanes x 14 planes of %d container(s) each, plus the bench's own frame / flow planes (every lane works "
                "on planes of its own)" % (job.n_lanes, job.group if job.step_group == 1 else job.step_group),
    }
    # <- the number: the timed region of exactly K steps, run `--repeats` times back to back after ONE warm-up; the line
    # reports the MEDIAN region (ms_per_step, value, pairs_per_s) with the fastest and slowest beside it
    regions = timed_region(job, batch, torch, args.steps, args.warmup, args.repeats)
    elapsed = float(np.median(regions))
    check = output_check(job)
    oracle_timing = None
    if rank == 0 and not args.no_oracle_check:
        check["oracle"] = oracle_check(job)
        oracle_timing = check["oracle"].pop("_timing")
        check["ok"] = check["ok"] and check["oracle"]["first_pair_equals_cpu_oracle"]
    # (its own leg after the timed region: at least 64 steps, so that four lanes' fill and drain do not weigh on a short --steps run)
    host_entry = host_entry_leg(job, batch, torch, max(64, args.steps)) if not args.no_host_entry_leg else None
    if host_entry is not None and not host_entry["flows_bit_identical_to_device_resident_run"]:
        check["ok"] = False
    first_pair = job.first_pair
    n_lanes, step_group = job.n_lanes, job.step_group
    levels_run = int(min(cfg["levels"], flow2d.host_lib().flow2d_host_max_warp_level_static(w, h, cfg["scale"])))
    # what the sampling legs below need of the job, which is closed before the batch leg (the legs make contexts of their own)
    import types
    sample = types.SimpleNamespace(flow2d=flow2d, cfg=cfg, local_rank=local_rank, first_pair=first_pair, params=job.params,
                                   sync=lambda: None)
    job.close()

    if not args.no_batch_leg and args.workload != BATCH_WORKLOAD and args.batch_leg_last:  # (developer A/B: where rounds 4-5 had it)
        batch_result = batch_leg(flow2d, batch, torch, args, rank, local_rank, world)

    finest, pair_latency_ms = roofline_sample(sample)
    sweep_ms = per_sweep_sample(sample) if rank == 0 else None
    copy_gbs = measured_copy_peak(flow2d, local_rank) if rank == 0 else None

    ok = check["ok"] and (batch_result is None or batch_result["output_check"]["ok"])
    if rank == 0:
        pairs_total = args.steps * cfg["pairs_per_rank"] * world
        px_iters = float(w) * h * cfg["outer"] * cfg["inner"]
        solve_ms = float(np.mean([r[2] for r in finest]))
        launch_ms = [r[3] / r[4] for r in finest]
        kernel_ms = float(np.mean(launch_ms))
        launches = finest[-1][4]
        bytes_per_launch = float(finest[-1][5])
        algorithm_used = int(finest[-1][6])  # from the timing record: what flow2d_solve_level actually ran
        algorithmic = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        pmc_run = pmc_select(pmc_run, cfg, algorithm_used)
        if pmc_run.get("hbm_bytes_per_launch"):
            pmc, pmc_source = pmc_run, ("this run: rocprofv3 --kernel-trace --pmc child passes of the same workload "
                                        "(FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVES), before the timed region")
        else:
            pmc = load_pmc(args.workload, algorithm_used)
            pmc_source = ("profiles/traffic.json (OFFLINE: measured by tools/pmc_passes.sh at commit %s, not in this run%s)"
                          % (pmc.get("measured_at_commit", "9b4994d"),
                             "; in-run passes failed: " + pmc_run["error"] if pmc_run.get("error") else "")) if pmc else None
        phys = pmc.get("hbm_bytes_per_launch")
        valu = pmc.get("valu_insts_per_launch")
        phys_gbs = phys / (kernel_ms * 1e-3) / 1e9 if phys else None
        valu_frac = valu * VALU_ISSUE_CYCLES / (SIMDS * CLOCK_HZ * kernel_ms * 1e-3) if valu else None
        temporal = algorithm_used in (2, 3, 4)  # several sweeps per trip through HBM: not bound by the per-sweep bytes
        # the launch's distance to each bound.  Clock: per XCD, held during the sampled passes (above).
        clocks = [g for g in getattr(sample, "sample_clock_ghz", []) if g > 0]
        clock_ghz = float(np.mean(clocks)) if clocks else None
        compulsory = float(w) * h * 4 * 8  # a fused pass reads f0, f1, u, v, du, dv and writes du, dv: eight planes once
        hbm_floor_us = compulsory / (copy_gbs * 1e9) * 1e6 if copy_gbs else None
        probes = probe_builds(args, cfg) if rank == 0 and algorithm_used == 2 and not args.no_probe_builds else {}
        kernel_name = {1: "Jacobi sweep kernel (%s)" % {0: "solve_2d", 1: "solve_2d_grad", 3: "solve_2d_log"}.get(
                           cfg["constancy"], "gradient-untiled"),
                       2: ("fused outer-iteration strip kernel (phi/ksi + %d red-black SOR iterations per launch)" % min(cfg["inner"], 2))
                          if cfg.get("sor_omega") else
                          "fused outer-iteration strip kernel (phi/ksi + %d Jacobi sweeps per launch)" % min(cfg["inner"], 5),
                       3: "single-workgroup level kernel", 4: "tiled outer-iteration kernel (LDS tiles)"}[algorithm_used]
        roof = {
            # what bounds the kernel: vector-ALU instruction issue for the kernels that keep the sweeps of an outer
            # iteration on chip, HBM for the per-sweep kernels.  achieved / frac = PHYSICAL HBM bytes (PMC) over the measured
            # launch time against the 8 TB/s peak; effective_* = the contract's ALGORITHMIC bytes (the
            # reference's per-sweep schedule, 32 + 40 x inner B per pixel and outer iteration) over the same time, which
            # exceeds the peak for a temporally blocked kernel because it never moves those bytes.
            "bound": "valu" if temporal else "hbm",
            "kernel": kernel_name,
            "achieved": round(phys_gbs, 1) if phys_gbs else None,
            "peak": HBM_PEAK_GBS,
            "measured_copy_gbs": copy_gbs,  # a 512 MiB device-to-device copy on this box (read + write), for scale
            "unit": "GB/s",
            "frac": round(phys_gbs / HBM_PEAK_GBS, 4) if phys_gbs else None,
            "traffic": phys,
            "effective_achieved": round(algorithmic, 1),
            "effective_frac": round(algorithmic / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "traffic_over_algorithmic": round(phys / bytes_per_launch, 4) if phys else None,
            "avg_launch_ms": round(kernel_ms, 5),
            "launch_samples": len(finest) * launches,
            "launches_per_level_solve": launches,
            "traffic_first_launch_of_level": pmc.get("hbm_bytes_first_launch"),
            "valu_instr_per_launch": valu,
            "valu_issue_frac": round(valu_frac, 4) if valu_frac else None,  # of one wave64 instruction per 2 cycles per SIMD
            "valu_peak_instr_per_s": SIMDS * CLOCK_HZ / VALU_ISSUE_CYCLES,
            # ... and at the clock the chip held while these launches ran (round 6): the shader clock is not 2.4 GHz under this
            # kernel -- power management holds 1.6-2.1 GHz, per XCD 3-5 % apart -- so the nominal fraction mixes DVFS into issue efficiency
            "shader_clock_ghz": round(clock_ghz, 3) if clock_ghz else None,
            "shader_clock_ghz_per_xcd": getattr(sample, "sample_clock_ghz", None),
            "valu_issue_frac_at_clock": round(valu * VALU_ISSUE_CYCLES / (SIMDS * clock_ghz * 1e9 * kernel_ms * 1e-3), 4) if valu and clock_ghz else None,
            # the launch's floor on the memory side: the eight planes a fused pass must move once, at this box's plain streaming rate
            "compulsory_bytes_per_launch": compulsory if temporal else None,
            "hbm_floor_us": round(hbm_floor_us, 1) if hbm_floor_us and temporal else None,
            # timing probes of the same kernel (developer builds under ab/, when present): its loads and stores without the arithmetic,
            # its arithmetic on cache-resident rows -- each at the clock IT holds (memory-only ~2.35 GHz, compute-only 2.0-2.3, the
            # whole kernel 1.6-2.1: the same cycle count per wave with and without the HBM traffic, profiles/r06_experiments)
            "memory_only_us": probes.get("memory_only_us"),
            "compute_only_us": probes.get("compute_only_us"),
            # SIMD cycles (at the nominal 2.4 GHz; the kernel holds 2.05-2.3) per wave64 VALU instruction of the launch: two
            # waves of this kernel's instruction mix on a SIMD get through one per about 4.3 cycles when both are busy
            # (per-wave stamps, profiles/r04_experiments/README.md sections 2-3) -- that, not one per 2 cycles, is its ceiling
            "valu_cycles_per_instr_per_simd": round(SIMDS * CLOCK_HZ * kernel_ms * 1e-3 / valu, 2) if valu else None,
            # SURVEY 8(d) metric 1, second half -- "the pure inner-sweep rate (events around K7 only)": the per-sweep Jacobi
            # kernel (one launch per reference launch of solve_2d*) at this workload's finest level, alone on the GPU.  It is
            # the path of inner = 1, red-black SOR and planes of 4 GiB and more, and the literal subject of north_star's
            # "70 % of the HBM roofline on the SOR inner loop"; the pyramids of this line run the fused kernel above instead.
            "per_sweep": per_sweep_block(cfg, sweep_ms, pmc.get("per_sweep") or {}),
            "pmc_source": pmc_source,
            "pmc_detail": {k: pmc_run[k] for k in ("kernel", "fetch_size_correction", "write_size_correction",
                                                    "correction_from", "launches_sampled", "waves_per_launch")
                           if k in pmc_run} or None,
            "note": ("frac = physical HBM fraction; effective_frac = algorithmic bytes of the reference's per-sweep schedule "
                     "over the launch time (> 1: temporal blocking); the kernel is bound by VALU issue at 2 waves/SIMD "
                     "(valu_issue_frac of the issue peak; DESIGN.md 3.1)") if temporal else
                    "frac = physical HBM fraction; effective_frac = algorithmic bytes over the launch time",
        }
        out = {
            "metric": "Mpixels*SOR-iters/sec at finest level (whole-pyramid wall time); full-pyramid pairs/sec in pairs_per_s",
            "value": round(px_iters * pairs_total / elapsed / 1e6, 1),
            "unit": "Mpixel*iters/s",
            "n_gpus": world,
            **({"rehearsal": "%d ranks sharing ONE GPU over gloo (--rehearse-on-one-gpu): a walk through the multi-rank "
                             "branches, not a measurement" % world} if args.rehearse_on_one_gpu else {}),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_min": round(min(regions) / args.steps * 1e3, 4),
            "ms_per_step_max": round(max(regions) / args.steps * 1e3, 4),
            "timed_regions": len(regions),  # each of exactly `steps` steps; ms_per_step is their median
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if cfg.get("frames") != "rub" else "rub1 / rub2, the reference's sample pair (tests/data)",
            "config": {
                "workload": args.workload, "width": w, "height": h, "pairs_per_gpu_per_step": cfg["pairs_per_rank"],
                "pyramid_levels": levels_run,
                "warp_scale": cfg["scale"], "outer_iterations": cfg["outer"], "inner_iterations": cfg["inner"],
                "data_constancy": CONSTANCY_NAME[cfg["constancy"]], "median_radius": cfg["median"],
                "gaussian_sigma": cfg["sigma"], "alpha": cfg["alpha"], "solver_algorithm": algorithm_used,
                "relaxation": ("red-black SOR, omega %g (opt-in; no reference parity: the reference is Jacobi)" % cfg["sor_omega"])
                              if cfg.get("sor_omega") else "Jacobi, reference iteration counts (bit-exact parity mode)",
                "parallelism": "independent pairs, one process per GPU, no data-path collective",
                "host_path": "OpticalFlowBatch2D::ComputeFlowBatchDevice (C++): one call per step",
                "streams_per_gpu": n_lanes, "hip_graph_replay": not args.no_graph,
                "steps_per_lock_step_group": step_group,
                "group_planes": None if step_group <= 1 else
                                ("every plane an allocation of its own: gathered and handed back by the object (two device copies per group)"
                                 if args.scattered_groups else
                                 "a lane's pairs one container apart in four allocations (frame 0, frame 1, u, v): a group as laid out, run in place"),
                "batch_mode": ("lock-step group" if args.batch_mode == "groups" else "lanes") if cfg["pairs_per_rank"] > 1 else None,
                "timed_region": "graph-replayed steps only; output check, roofline sample, batch leg and baselines follow it",
            },
            "pairs_per_s": round(pairs_total / elapsed, 3),  # inputs and outputs resident in HBM (the bench contract)
            # SURVEY 8(d) metric 2 as defined: host images in, host flows out, H<->D inside the bracket
            "pairs_per_s_incl_h2d": host_entry["pairs_per_s"] if host_entry else None,
            "host_entry": host_entry,
            "single_pair_latency_ms": round(pair_latency_ms, 3),  # one pair alone on the GPU, graph replay, launch to done
            "pairs_per_s_single": round(1e3 / pair_latency_ms, 3),  # = 1 / single_pair_latency: no second pair in flight
            # the same lone pair with the pipeline's kernels (round 5's single pair; OpticalFlow2D::lone = false)
            "single_pair_latency_pipeline_kernels_ms": round(getattr(sample, "single_stream_latency_ms", float("nan")), 3),
            "finest_level": {
                "solve_ms": round(solve_ms, 4),
                "mpix_iters_per_s": round(px_iters / (solve_ms * 1e-3) / 1e6, 1),
                "algorithmic_gbs": round(w * h * cfg["outer"] * (32 + 40 * cfg["inner"]) / (solve_ms * 1e-3) / 1e9, 1),
                "sampled": "eager passes after the timed region, alone on the GPU",
            },
            "roofline": roof,
            "output_check": check,
            "batch": batch_result,
            "device_memory": dict({"used_gib": round((total_b - free_b) / 2 ** 30, 3), "total_gib": round(total_b / 2 ** 30, 1)},
                                  **memory_parts),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, full_run=oracle_timing)
        else:
            out["cpu_baseline"] = None
        if cfg.get("sor_omega"):
            # the algorithm north_star names, beside the reference's Jacobi: iterations and milliseconds to the same residual
            out["sor_time_to_residual"] = sor_time_to_residual(flow2d, cfg, local_rank, first_pair)
        if world == 1 and not args.no_reference_baseline and not cfg.get("sor_omega"):  # (the reference has no SOR)
            ref = out["reference_gpu_baseline"] = reference_gpu_baseline(cfg, *first_pair)
            if ref and ref.get("pairs_per_s") and host_entry:  # like for like: both brackets hold the H<->D copies
                ref["product_incl_h2d_over_reference"] = round(host_entry["pairs_per_s"] / ref["pairs_per_s"], 2)
        else:
            out["reference_gpu_baseline"] = None
        print(json.dumps(out))
    batch.shutdown()
    if not ok:
        sys.exit("bench.py: output check failed (graph replay / eager recomputation / CPU oracle / host entry disagree)")


if __name__ == "__main__":
    main()
