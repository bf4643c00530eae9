"""Summarise a rocprofv3 --pmc counter_collection.csv: one line per (kernel, grid) with mean counter values.
python tools/pmc_summary.py <csv> [name-filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else "conv"
agg = collections.OrderedDict()
for r in rows:
    if flt not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    k = (name, r["Grid_Size"], r["Dispatch_Id"])
    agg.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    agg[k]["_dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
by = collections.OrderedDict()
for (name, grid, _), c in agg.items():
    by.setdefault((name, grid), []).append(c)
for (name, grid), cs in by.items():
    cs = cs[len(cs) // 2:]  # later dispatches (warm)
    mean = {k: sum(c[k] for c in cs) / len(cs) for k in cs[0]}
    wc = mean.get("SQ_WAVE_CYCLES", 0.0)
    parts = [f"{name[:34]:34s} grid {int(grid):8d} dur_us {mean['_dur'] / 1e3:8.1f}"]
    for k, v in mean.items():
        if k in ("_dur",):
            continue
        if k.startswith("SQ_WAIT") or k.startswith("SQ_ACTIVE") and wc:
            parts.append(f"{k[3:]} {v / wc:5.2f}")
        else:
            parts.append(f"{k[3:] if k.startswith('SQ_') else k} {v:.3g}")
    print("  ".join(parts))
