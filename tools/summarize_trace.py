"""Per-kernel, per-grid-size summary of a rocprofv3 --kernel-trace CSV (developer tool).
usage: python tools/summarize_trace.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    d = collections.defaultdict(list)
    for r in rows:
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]) * 100000 + int(r.get("Grid_Size_Z", 1) or 1), int(r["VGPR_Count"]),
               int(r["LDS_Block_Size"]))
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in d.values())
    print("%-44s %14s %6s %5s %7s %10s %10s %6s" % ("kernel", "grid(threads)", "vgpr", "lds", "calls", "avg_us",
                                                   "total_ms", "%"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print("%-44s %14s %6d %5d %7d %10.2f %10.3f %6.2f" % (k[0][:44], "%dx%dx%d" % (k[1], k[2] // 100000, k[2] % 100000), k[3], k[4], len(v),
                                                            sum(v) / len(v) / 1e3, sum(v) / 1e6,
                                                            100.0 * sum(v) / total))


if __name__ == "__main__":
    main()
