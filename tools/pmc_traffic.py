"""Turns the rocprofv3 --pmc passes of tools/pmc_workload.py into profiles/traffic.json.

HBM bytes per launch = FETCH_SIZE * read_correction + WRITE_SIZE (both reported in KiB), where
read_correction is measured in the same run on two calibration kernels with exactly known traffic
(add_2d: 16 B per lane, 1-tap Gaussian rows: 4 B per lane).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
counts 128-byte requests as 64 bytes (factor 2 for wide coalesced reads); WRITE_SIZE is exact.
usage: python tools/pmc_traffic.py gpurun_out [size]
"""
import collections
import csv
import glob
import json
import os
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    root = sys.argv[1]
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    plane_kib = size * size * 4 / 1024.0
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "pmc_*", "*", "*counter_collection.csv")):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            res[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in res.items()}
    # fused kernel: every 4th launch (tools/pmc_workload.py runs 4 outer iterations per level solve) is the first
    # launch of a level, which starts from du = dv = 0 without reading the planes; the other three are steady state
    first = {k: {n: sum(v[0::4]) / len(v[0::4]) for n, v in c.items()} for k, c in res.items() if k.startswith("fused")}
    steady = {k: {n: sum(x for i, x in enumerate(v) if i % 4) / len([1 for i in range(len(v)) if i % 4])
                  for n, v in c.items()} for k, c in res.items() if k.startswith("fused")}
    corr16 = 2 * plane_kib / avg["add_2d_kernel"]["FETCH_SIZE"]
    corr4 = plane_kib / avg["gauss_kernel<true>"]["FETCH_SIZE"]
    wr16 = plane_kib / avg["add_2d_kernel"]["WRITE_SIZE"]
    out = {"_calibration": {
        "size": size,
        "fetch_size_correction_16B_per_lane": round(corr16, 4),
        "fetch_size_correction_4B_per_lane": round(corr4, 4),
        "write_size_correction": round(wr16, 4),
        "note": "FETCH_SIZE/WRITE_SIZE in KiB from separate --pmc passes; corrections from add_2d / 1-tap gauss with known bytes",
    }}
    px = float(size) * size
    table = {
        "fused_outer_kernel<5, 0, true, false>": ("cfg3_4096_grey/algorithm2", px * (32 + 40 * 5)),
        "fused_outer_kernel<5, 1, true, false>": ("cfg3_4096_gradient/algorithm2", px * (32 + 40 * 5)),
        "sweep_grey_kernel": ("cfg3_4096_grey/algorithm1", px * 40),
        "sweep_grad_kernel": ("cfg3_4096_gradient/algorithm1", px * 40),
        "phi_ksi_kernel": ("phi_ksi_4096", px * 32),
    }
    for kernel, (key, algorithmic) in table.items():
        if kernel not in avg:
            continue
        src = steady.get(kernel, avg[kernel])  # fused: steady-state launches
        rd = src["FETCH_SIZE"] * corr4 * 1024
        wr = src["WRITE_SIZE"] * wr16 * 1024
        hit, miss = src.get("TCC_HIT_sum"), src.get("TCC_MISS_sum")
        out[key] = {
            "kernel": kernel,
            "hbm_read_bytes_per_launch": round(rd),
            "hbm_write_bytes_per_launch": round(wr),
            "hbm_bytes_per_launch": round(rd + wr),
            "algorithmic_bytes_per_launch": round(algorithmic),
            "traffic_over_algorithmic": round((rd + wr) / algorithmic, 4),
            "l2_hit_rate": round(hit / (hit + miss), 4) if hit is not None and miss is not None else None,
            "valu_insts_per_launch": round(src["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in src else None,
            "waves_per_launch": round(src["SQ_WAVES"]) if "SQ_WAVES" in src else None,
        }
        if kernel in first:
            out[key]["hbm_bytes_first_launch"] = round(first[kernel]["FETCH_SIZE"] * corr4 * 1024 +
                                                       first[kernel]["WRITE_SIZE"] * wr16 * 1024)
            out[key]["note"] = "per-launch figures are steady-state launches (outer iterations 2..n of a level)"
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
