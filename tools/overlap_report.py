"""How busy the GPU is in a pipelined bench run, from a rocprofv3 --kernel-trace CSV (developer tool).
Prints wall time, the union of kernel intervals (time with at least one kernel running), the sum of kernel
durations, and per kernel/grid the share of its own time during which another kernel ran too.
usage: python tools/overlap_report.py <kernel_trace.csv> [skip_first_fraction]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    ev = []
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        grid = "%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name + " " + grid))
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    cut = t0 + (t1 - t0) * skip  # drop warm-up / capture
    ev = [e for e in ev if e[0] >= cut]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    # sweep line
    points = []
    for i, (s, e, _) in enumerate(ev):
        points.append((s, 1, i))
        points.append((e, -1, i))
    points.sort()
    active = set()
    busy = 0
    multi = collections.Counter()
    own = collections.Counter()
    last = points[0][0]
    for t, kind, i in points:
        dt = t - last
        if dt > 0 and active:
            busy += dt
            for j in active:
                own[ev[j][2]] += dt
                if len(active) > 1:
                    multi[ev[j][2]] += dt
        last = t
        if kind == 1:
            active.add(i)
        else:
            active.discard(i)
    total = sum(e - s for s, e, _ in ev)
    print("wall %.3f ms, busy (>=1 kernel) %.3f ms = %.1f %%, sum of kernel durations %.3f ms (avg concurrency %.2f)" %
          ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), total / 1e6, total / busy))
    print("%-60s %10s %10s" % ("kernel grid", "own ms", "overlapped"))
    for k, v in own.most_common(14):
        print("%-60s %10.3f %9.1f%%" % (k[:60], v / 1e6, 100.0 * multi[k] / v))


if __name__ == "__main__":
    main()
