"""Developer tool (round 4): does the row pitch of the containers change the fused strips' launch time?

Times a 10 x 5 level solve with the fused strips (and the per-sweep kernels) for a list of cases
    WxH[@container_width][*instances]
e.g. 4096x4096 4096x4096@4160 1920x1080*8 1920x1080@2048*8 -- the level lives in the top-left corner of containers
`container_width` wide (pitch = container_width * 4 rounded up to 256 B), `instances` containers one below the other
as a lock-step group (flow2d_context_set_batch).  Prints the median and the minimum over the repetitions, per launch
(level solve / 10).  usage (GPU box): python tools/pitch_ab.py [cases...]"""
import importlib
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = importlib.import_module("cuda-flow2d_amd")

DEFAULT = ["4096x4096", "4096x4096@4160", "4096x4096@4128", "4096x4096@4224", "4032x4096", "4160x4096",
           "1920x1080*8", "1920x1080@2048*8", "1920x8640", "1920x8640@2048", "2048x2048", "2048x2048@2112"]


def run(case, algos, reps):
    inst = 1
    if "*" in case:
        case, n = case.split("*")
        inst = int(n)
    cw = None
    if "@" in case:
        case, c = case.split("@")
        cw = int(c)
    w, h = (int(t) for t in case.split("x"))
    cw = cw or w
    ctx = F.Context(0)
    rng = np.random.default_rng(0)
    tall = h * inst
    if "--smooth" in sys.argv:  # the bench's kind of data: SURVEY 8(d) frames, a flow near (2, 0)
        y, x = np.mgrid[0:tall, 0:cw].astype(np.float64)
        img = lambda xx, yy: (128.0 + 60.0 * np.sin(2 * np.pi * xx / 64.0) * np.cos(2 * np.pi * yy / 48.0)
                              + 30.0 * np.sin(2 * np.pi * (xx + 2 * yy) / 23.7)).astype(np.float32)
        datas = [img(x, y), img(x - 0.03, y + 0.02), (2.0 + 0.01 * np.sin(x / 50.0)).astype(np.float32),
                 (0.01 * np.cos(y / 70.0)).astype(np.float32)]
    elif "--zeros" in sys.argv:
        datas = [np.zeros((tall, cw), np.float32)] * 4
    else:
        datas = [rng.normal(0, 1, (tall, cw)).astype(np.float32)] * 4
    planes = [ctx.plane(cw, tall, d) for d in datas]
    du, dv, phi, ksi, tdu, tdv = (ctx.plane(cw, tall).fill_bytes(0) for _ in range(6))
    pitch = planes[0].pitch
    if inst > 1:
        import ctypes
        F.hip_lib().flow2d_context_set_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
        rc = F.hip_lib().flow2d_context_set_batch(ctx.handle, inst, pitch * h)
        assert rc == 0, rc
    line = "%5dx%-5d x%d pitch %6d" % (w, h, inst, pitch)
    for constancy in (0, 1):
        for algo in algos:
            ts = []
            for rep in range(reps):
                e0, e1 = ctx.event(), ctx.event()
                ctx.record(e0)
                ctx.solve_level(*planes, du, dv, phi, ksi, tdu, tdv, w, h, 1.0, 1.0, 35.0, 0.001, 0.001, 10, 5, constancy,
                                algo, container_height=h)
                ctx.record(e1)
                ts.append(ctx.elapsed_ms(e0, e1))
            ts = ts[reps // 3:]
            line += "  %s/%s med %.1f min %.1f us" % ("grey" if constancy == 0 else "grad",
                                                      {F.SOLVER_FUSED: "strips", F.SOLVER_PER_SWEEP: "sweeps"}[algo],
                                                      statistics.median(ts) * 100, min(ts) * 100)
    n = __import__("ctypes").c_ulonglong(0)
    F.hip_lib().flow2d_fused_fallbacks(ctx.handle, __import__("ctypes").byref(n))
    print(line + "  fallbacks %d" % n.value, flush=True)
    if inst > 1:
        F.hip_lib().flow2d_context_set_batch(ctx.handle, 1, 0)
    ctx.close()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    algos = [F.SOLVER_FUSED] + ([F.SOLVER_PER_SWEEP] if "--sweeps" in sys.argv else [])
    for case in (args or DEFAULT):
        run(case, algos, 6 if "--once" in sys.argv else 24)


if __name__ == "__main__":
    main()
