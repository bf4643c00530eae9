"""Diagnostic: from a rocprofv3 kernel-trace csv, the busy / gap split of the last N kernel dispatches."""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
busy = sum(e - s for s, e in zip(st, en))
gaps = sum(max(0, st[i + 1] - en[i]) for i in range(len(rows) - 1))
span = en[-1] - st[0]
per = collections.defaultdict(lambda: [0, 0])
for r, s, e in zip(rows, st, en):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    per[k][0] += e - s
    per[k][1] += 1
top = sorted(per.items(), key=lambda kv: -kv[1][0])[:10]
print(f"span {span / 1e6:.2f} ms busy {busy / 1e6:.2f} gaps {gaps / 1e6:.2f} ({len(rows)} kernels) | " +
      " ".join(f"{k}:{v[0] / v[1] / 1e3:.1f}us*{v[1]}" for k, v in top))
