"""Developer tool (round 4): where does a launch of the fused strip kernel spend its time, wave by wave?

Needs a developer build of the library with the wave stamps compiled in:
    make -C cuda-flow2d_amd/csrc BUILD=build_stamps LIB=$PWD/ab/stamps.so EXTRA="-DFLOW2D_DEV_BUILD -DFLOW2D_FUSED_STAMPS -DFLOW2D_FUSED_DEV"   (add -j8)
    FLOW2D_HIP_LIB=$PWD/ab/stamps.so python tools/fused_wave_stamps.py [WxH[*instances]] [grey|grad]
Every wave records its start and end on the 100 MHz clock, its shader cycles, where it ran (XCC, SE, CU, SIMD) and which
strip it had.  Printed per launch: the span of the launch, the distribution of the waves' lifetimes, the spread of their
start and end times, the clock they held, and the lifetimes by strip kind, by XCC and by how many waves shared the SIMD."""
import collections
import ctypes as C
import importlib
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")


STALL_WAVES = 1 << 12


def fetch(lib, with_stalls=False):
    """the stamps recorded since the last call; with_stalls: also the stall histograms [waves][3][64] of the first 4096 of them"""
    n = C.c_size_t(0)
    buf = np.zeros((1 << 16, 8), np.uint64)
    stalls = np.zeros((STALL_WAVES, 3, 64), np.uint32)
    rc = lib.flow2d_dev_fused_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(1 << 16), C.byref(n),
                                     stalls.ctypes.data_as(C.c_void_p) if with_stalls else None)
    assert rc == 0, rc
    if with_stalls:
        return buf[:n.value], stalls[:min(n.value, STALL_WAVES)]
    return buf[:n.value]


def report_stalls(st, stalls, label):
    """Where the waves of ONE launch waited at the row commit (s_waitcnt for the row's loads between two clock readings; a reading
    pair costs ~50 cycles when nothing is outstanding -- the floor of the histogram)."""
    n = len(stalls)
    st = st[:n]
    cyc = st[:, 2].astype(np.float64)
    hw = st[:, 3].astype(np.int64)
    xcc = st[:, 4].astype(np.int64) & 15
    cnt = stalls[:, 0, :32].astype(np.float64)
    cyc_bin = stalls[:, 1, :32].astype(np.float64)
    prog = stalls[:, 2, :].astype(np.float64)
    total = cyc_bin.sum()
    print("%s: stalls of %d waves: %.0f waits per wave, %.1f %% of the waves' cycles at the row commit (%.0f of %.0f cycles per wave)" %
          (label, n, cnt.sum() / n, 100 * total / cyc.sum(), total / n, cyc.mean()))
    print("  by duration   cycles   waits/wave   share of stall cycles   share of wave cycles")
    for b in range(32):
        if cnt[:, b].sum() > 0:
            print("   2^%-2d %7d..%-7d %9.2f %12.1f %% %12.2f %%" % (b, 1 << b, (2 << b) - 1, cnt[:, b].sum() / n,
                                                                 100 * cyc_bin[:, b].sum() / total, 100 * cyc_bin[:, b].sum() / cyc.sum()))
    # the floor: a wait that finds its row there costs the two clock readings; everything above ~2x the modal bin is real waiting
    per_wave = cyc_bin.sum(axis=1) / cyc
    print("  per wave, stall share of its cycles: min %.1f %%  p10 %.1f  median %.1f  p90 %.1f  max %.1f" %
          (100 * per_wave.min(), 100 * pct(per_wave, 10), 100 * pct(per_wave, 50), 100 * pct(per_wave, 90), 100 * per_wave.max()))
    slot = hw & 15
    print("  by wave slot on the SIMD (waves, median stall share %%): " +
          "  ".join("%d: %d %.1f" % (k, (slot == k).sum(), 100 * pct(per_wave[slot == k], 50)) for k in sorted(set(slot.tolist()))))
    print("  by XCC (median stall share %%): " + "  ".join("%d: %.1f" % (k, 100 * pct(per_wave[xcc == k], 50)) for k in sorted(set(xcc.tolist()))))
    steps = prog.sum(axis=0) / n
    used = np.nonzero(steps)[0]
    print("  stall cycles per row step by the wave's progress (four steps per column, mean over waves):")
    print("   " + " ".join("%4.0f" % (steps[k] / 4) for k in range(used.max() + 1)))


def pct(v, q):
    return float(np.percentile(v, q))


def report(st, label):
    t0 = st[:, 0].astype(np.int64) * 10  # ns
    t1 = st[:, 1].astype(np.int64) * 10
    cyc = st[:, 2].astype(np.float64)
    hw = st[:, 3].astype(np.int64)
    xcc = st[:, 4].astype(np.int64) & 15
    edge = (st[:, 6].astype(np.int64) >> 8) & 1
    rows = (st[:, 7].astype(np.int64) >> 32) - (st[:, 7].astype(np.int64) & 0xffffffff)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    life = (t1 - t0) / 1e3  # us
    base = t0.min()
    span = (t1.max() - base) / 1e3
    print("%s: %d waves, launch span %.1f us" % (label, len(st), span))
    print("  wave lifetime us: min %.1f  p10 %.1f  median %.1f  mean %.1f  p90 %.1f  max %.1f" %
          (life.min(), pct(life, 10), pct(life, 50), life.mean(), pct(life, 90), life.max()))
    print("  start after launch begin us: median %.1f  p90 %.1f  max %.1f;   end before launch end us: median %.1f  p90 %.1f  max %.1f" %
          (pct((t0 - base) / 1e3, 50), pct((t0 - base) / 1e3, 90), (t0 - base).max() / 1e3,
           pct(span - (t1 - base) / 1e3, 50), pct(span - (t1 - base) / 1e3, 90), (span - (t1 - base) / 1e3).max()))
    print("  shader clock held (cycles / lifetime): median %.2f GHz  min %.2f  max %.2f" %
          (pct(cyc / (life * 1e3), 50), (cyc / (life * 1e3)).min(), (cyc / (life * 1e3)).max()))
    for e in (0, 1):
        m = edge == e
        if m.any():
            steps = rows[m] + 13
            print("  %s strips: %5d waves, rows %d..%d, lifetime median %.1f us (%.3f us, %.0f cycles per row step)" %
                  ("border  " if e else "interior", m.sum(), rows[m].min(), rows[m].max(), pct(life[m], 50),
                   pct(life[m] / steps, 50), pct(cyc[m] / steps, 50)))
    print("  by XCC (waves, median lifetime, last end): " +
          "  ".join("%d: %d %.1f %.1f" % (x, (xcc == x).sum(), pct(life[xcc == x], 50), (t1[xcc == x].max() - base) / 1e3)
                    for x in sorted(set(xcc.tolist()))))
    # how many waves of the launch shared a SIMD
    key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
    cnt = collections.Counter(key.tolist())
    share = np.array([cnt[k] for k in key.tolist()])
    print("  SIMDs used %d; waves by SIMD occupancy: " % len(cnt) +
          "  ".join("%d per SIMD: %d waves, median lifetime %.1f" % (c, (share == c).sum(), pct(life[share == c], 50))
                    for c in sorted(set(share.tolist()))))
    slots = collections.defaultdict(list)
    for k, wv in zip(key.tolist(), (hw & 15).tolist()):
        slots[k].append(wv)
    print("  wave slots of the waves that shared a SIMD: %s" %
          dict(collections.Counter(tuple(sorted(v)) for v in slots.values()).most_common(6)))
    cus = collections.Counter((key // 4).tolist())
    print("  CUs used %d; waves per CU: %s" % (len(cus), dict(collections.Counter(cus.values()))))
    # The launch lasts as long as its last wave.  How far is that from the mean, and where does the spread come from: the XCDs' clocks
    # (cycles per wave are the work, cycles / lifetime the clock the wave saw), the SIMD a wave shared, its age on that SIMD?
    clock = cyc / (life * 1e3)
    end = (t1 - base) / 1e3
    print("  launch span %.1f = mean wave end %.1f + %.1f (%.1f %%);  per XCC: clock GHz / median cycles per wave (10^3) / median end / last end:" %
          (span, end.mean(), span - end.mean(), 100 * (span - end.mean()) / span))
    print("    " + "  ".join("%d: %.3f / %.0f / %.1f / %.1f" % (x, pct(clock[xcc == x], 50), pct(cyc[xcc == x], 50) / 1e3, pct(end[xcc == x], 50),
                                                              end[xcc == x].max()) for x in sorted(set(xcc.tolist()))))
    simd_end = collections.defaultdict(float)
    simd_cyc = collections.defaultdict(float)
    for k, e, c in zip(key.tolist(), end.tolist(), cyc.tolist()):
        simd_end[k] = max(simd_end[k], e)
        simd_cyc[k] += c
    se = np.array(list(simd_end.values()))
    print("  SIMD done (its last wave's end) us: p10 %.1f  median %.1f  p90 %.1f  max %.1f;  ends of the older / younger wave of a SIMD, median: %s" %
          (pct(se, 10), pct(se, 50), pct(se, 90), se.max(),
           " / ".join("%.1f" % pct(end[(hw & 15) == k], 50) for k in sorted(set((hw & 15).tolist())))))
    cu_end = collections.defaultdict(float)
    for k, e in zip((key // 4).tolist(), end.tolist()):
        cu_end[k] = max(cu_end[k], e)
    ce = np.array(list(cu_end.values()))
    print("  CU done us: p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (pct(ce, 10), pct(ce, 50), pct(ce, 90), ce.max()))


def main():
    args = [a for a in sys.argv[1:]]
    case = args[0] if args and "x" in args[0] else "4096x4096"
    modes = [a for a in args if a in ("grey", "grad")] or ["grey", "grad"]
    inst = 1
    if "*" in case:
        case, n = case.split("*")
        inst = int(n)
    w, h = (int(t) for t in case.split("x"))
    lib = F.hip_lib()
    if not hasattr(lib, "flow2d_dev_fused_stamps"):
        raise SystemExit("this library has no wave stamps: build ab/stamps.so and set FLOW2D_HIP_LIB (see the docstring)")
    ctx = F.Context(0)
    rng = np.random.default_rng(0)
    tall = h * inst
    planes = [ctx.plane(w, tall, rng.normal(0, 1, (tall, w)).astype(np.float32)) for _ in range(4)]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(w, tall).fill_bytes(0) for _ in range(6))
    if inst > 1:
        lib.flow2d_context_set_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        assert lib.flow2d_context_set_batch(ctx.handle, inst, planes[0].pitch * h) == 0
    for mode in modes:
        constancy = 0 if mode == "grey" else 1
        for rep in range(12):  # warm: sustained clocks
            ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 10, 5, constancy,
                            F.SOLVER_FUSED, container_height=h)
        ctx.synchronize()
        fetch(lib)
        with_stalls = os.environ.get("FLOW2D_STALLS", "1") != "0"
        e0, e1 = ctx.event(), ctx.event()
        # FLOW2D_STAMPS_SUSTAIN=n: n level solves of ten launches queued back to back before the stamped one -- the clock a launch
        # holds under SUSTAINED load (the ring keeps the last 32 launches; the trend over them is printed)
        sustain = int(os.environ.get("FLOW2D_STAMPS_SUSTAIN", "0"))
        for rep in range(sustain):
            ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 10, 5, constancy,
                            F.SOLVER_FUSED, container_height=h)
        ctx.record(e0)
        ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 4, 5, constancy,
                        F.SOLVER_FUSED, container_height=h)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)
        st, stalls = fetch(lib, True)
        # (the histograms belong to the first 4096 stamps in buffer order: keep that order's index through the sort below)
        index = np.arange(len(st))
        # split into launches by start time: the waves of one launch start within a few us, launches are >100 us apart
        order = np.argsort(st[:, 0])
        st, index = st[order], index[order]
        gaps = np.nonzero(np.diff(st[:, 0].astype(np.int64)) > 2000)[0]  # > 20 us
        launches = np.split(st, gaps + 1)
        launch_index = np.split(index, gaps + 1)
        print("== %dx%d x%d %s: 4 outer iterations %.1f us (%.1f per launch by events); %d launches found" %
              (w, h, inst, mode, ms * 1e3, ms * 1e3 / 4, len(launches)))
        if sustain:
            print("  per launch (oldest first): span us / median clock GHz / median cycles per interior wave (10^3):")
            rows = []
            for l in launches:
                lt0, lt1 = l[:, 0].astype(np.int64) * 10, l[:, 1].astype(np.int64) * 10
                life = (lt1 - lt0) / 1e3
                inner = ((l[:, 6].astype(np.int64) >> 8) & 1) == 0
                rows.append("%.1f/%.3f/%.0f" % ((lt1.max() - lt0.min()) / 1e3, pct(l[:, 2] / (life * 1e3), 50), pct(l[inner, 2], 50) / 1e3))
            print("   " + "  ".join(rows))
        out = os.environ.get("FLOW2D_STAMPS_OUT")
        if out:  # raw stamps of the last launch, for offline analysis
            np.save("%s_%dx%dx%d_%s.npy" % (out, w, h, inst, mode), launches[-1])
        for k, l in enumerate(launches):
            if k in (0, len(launches) - 1):
                report(l, "  launch %d%s" % (k, " (first of the level: du = dv = 0, not read)" if k == 0 else ""))
            if with_stalls and k == 1:  # the launches after the first hold the steady state; the histograms cover the first 4096 waves
                have = launch_index[k] < len(stalls)
                if have.sum() > 256:
                    report_stalls(l[have], stalls[launch_index[k][have]], "  launch %d" % k)
    ctx.close()


if __name__ == "__main__":
    main()
