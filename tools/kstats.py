"""Summarise a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> [steps]"""
import csv
import sys

f, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms/step {tot / 1e6 / steps:.3f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print(f"{r['Name'][:58]:58s} calls/step {int(r['Calls']) / steps:6.1f} ms/step {float(r['TotalDurationNs']) / 1e6 / steps:7.3f} "
          f"avg_us {float(r['AverageNs']) / 1e3:8.1f}")
