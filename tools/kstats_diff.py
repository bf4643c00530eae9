"""Per-kernel totals of two rocprofv3 kernel_stats.csv files side by side (ms per network evaluation: 126 evaluations = two
32-step solves of tools/sampler_profile.py), sorted by the first file's total:
    python tools/kstats_diff.py a_kernel_stats.csv b_kernel_stats.csv [evals=126]"""
import csv
import re
import sys


def load(p):
    out = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"^void ", "", n)
        n = n.split("(")[0][:58]
        t, c = out.get(n, (0.0, 0))
        out[n] = (t + float(r["TotalDurationNs"]) / 1e6, c + int(r["Calls"]))
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
ev = float(sys.argv[3]) if len(sys.argv) > 3 else 126.0
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0))[0] + b.get(n, (0, 0))[0]))
ta = tb = 0.0
print(f"{'kernel':58s} {'A ms/eval':>10s} {'calls':>6s} {'B ms/eval':>10s} {'calls':>6s} {'B-A':>8s}")
for n in names:
    xa, ca = a.get(n, (0.0, 0))
    xb, cb = b.get(n, (0.0, 0))
    ta += xa
    tb += xb
    if max(xa, xb) / ev >= 0.003:
        print(f"{n:58s} {xa / ev:10.3f} {ca:6d} {xb / ev:10.3f} {cb:6d} {(xb - xa) / ev:8.3f}")
print(f"{'TOTAL':58s} {ta / ev:10.3f} {'':6s} {tb / ev:10.3f} {'':6s} {(tb - ta) / ev:8.3f}")
