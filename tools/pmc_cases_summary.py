"""Developer tool: per-case means of the SQ / TCC counters collected by tools/pmc_cases.sh over
tools/pitch_ab.py --once --smooth CASE1 CASE2 (two cases): per kernel name the first 60 dispatches are case 1, the next
60 case 2 (both may launch the same number of threads, so the grid does not tell them apart); the first launch of a
level (no du, dv read) and the first three repetitions (cold) are left out.  usage: python tools/pmc_cases_summary.py pass*.csv"""
import collections
import csv
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    ids = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(path)) if "fused_outer" in r["Kernel_Name"]]
    short = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    for r in rows:
        if int(r["Dispatch_Id"]) not in ids[short(r)]:
            ids[short(r)].append(int(r["Dispatch_Id"]))
    for r in rows:
        pos = sorted(ids[short(r)]).index(int(r["Dispatch_Id"]))
        if pos % 10 == 0 or pos % 60 < 30:
            continue
        key = "%s  case %d" % (short(r), pos // 60 + 1)
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[key]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = sorted({c for cs in d.values() for c in cs})
keys = sorted(d)
print("%-30s" % "counter (mean per launch)" + "".join("%28s" % k.replace("fused_outer_kernel", "")[:27] for k in keys))
for c in names:
    print("%-30s" % c + "".join("%28.5g" % (sum(d[k][c]) / max(1, len(d[k][c]))) for k in keys))
print("derived:")
for k in keys:
    g = lambda c: sum(d[k][c]) / max(1, len(d[k][c]))
    print("  %-46s VALU active %.0f %%  issue wait %.0f %%  waitcnt %.0f %%  wave quad-cycles per wave %.0f  L2 hit %.0f %%  "
          "clock (GRBM_GUI_ACTIVE / 8 / ns) %.2f GHz" % (
              k, 100 * g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
              100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAVE_CYCLES") / g("SQ_WAVES"),
              100 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), g("GRBM_GUI_ACTIVE") / 8 / g("_ns")))
