"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (developer tool).
usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if want in name:
                res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(res.items()):
        print(k)
        for n, v in sorted(c.items()):
            print("   %-28s n=%-4d avg=%.4g" % (n, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
