"""Per-kernel time of ONE training step from a rocprofv3 kernel-trace csv of bench.py: steps are delimited by the
k_adam_ema launches; prints the median-span step's kernels grouped by name (ms per step, launches, average us).
    python tools/step_breakdown.py <kernel_trace.csv> [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_adam_ema" in r["Kernel_Name"]]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a + 1:b + 1]
    steps.append((int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"]), a, b))
steps.sort()
span, a, b = steps[len(steps) // 4]          # a fast-quartile step: inside the timed (replayed) region
seg = rows[a + 1:b + 1]
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*", "", k)[:60]
    per[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per[k][1] += 1
busy = sum(v[0] for v in per.values())
print(f"step span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, {len(seg)} launches")
for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{k:60s} {v[0] / 1e6:7.3f} ms  x{v[1]:3d}  avg {v[0] / v[1] / 1e3:7.1f} us")
