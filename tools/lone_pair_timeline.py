"""Developer tool (round 6): the LAST replay of tools/lone_pair_trace.py kernel by kernel from a rocprofv3 kernel trace:
queue, start and end relative to the replay's first kernel, duration, idle gap before it on its own queue, name and grid.
usage: python tools/lone_pair_timeline.py <kernel_trace.csv> [replays in the trace] [max rows]"""
import csv
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    # every replay launches the same kernels: the last replay is the last len / replays rows
    last = rows[len(rows) - len(rows) // replays:]
    t0 = int(last[0]["Start_Timestamp"])
    end = max(int(r["End_Timestamp"]) for r in last)
    queues = sorted({r["Queue_Id"] for r in last})
    print("last replay: %d kernels on %d queue(s), first start to last end %.1f us; busy time per queue: %s" %
          (len(last), len(queues), (end - t0) / 1e3,
           ", ".join("%s %.1f us" % (q, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last if r["Queue_Id"] == q) / 1e3) for q in queues)))
    prev_end = {}
    print("%5s %9s %9s %8s %8s  %s" % ("queue", "start", "end", "us", "gap", "kernel grid"))
    for r in last[:limit]:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = e
        print("%5s %9.1f %9.1f %8.1f %8.1f  %s %sx%sx%s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, short(r["Kernel_Name"]),
                                                         r["Grid_Size_X"], r["Grid_Size_Y"], r.get("Grid_Size_Z", "1")))


if __name__ == "__main__":
    main()
