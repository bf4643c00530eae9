#!/bin/bash
# Matrix-pipe busy share per kernel of the replayed training step (its own rocprofv3 --pmc pass: counters only):
#   tools/pmc_mfma_busy.sh -> gpurun_out/r05_pmc_mfma_busy.txt
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=$R/gpurun_out
rm -rf $O/pmc_busy
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_busy -- python3 $R/bench.py --steps 3 --warmup 2 --no-sampler --no-cpu-baseline --step-launch graph > /dev/null 2> $O/pmc_busy.log
python3 - $(ls $O/pmc_busy/*/*counter_collection.csv | head -1) > $O/r05_pmc_mfma_busy.txt <<'PY'
import collections, csv, sys
acc = collections.defaultdict(lambda: [0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        acc[name][0] += float(r["Counter_Value"])
    elif r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
        acc[name][1] += float(r["Counter_Value"])
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python3 bench.py --steps 3 --warmup 2 --no-sampler --no-cpu-baseline --step-launch graph")
print("# (its own pass: counters only, no trace domains).  Sums over every launch of a kernel in the run.  SQ_VALU_MFMA_BUSY_CYCLES counts per")
print("# SIMD (four per CU), SQ_BUSY_CU_CYCLES per CU: ratio / 4 = share of the CU-busy cycles in which a SIMD's matrix pipe was busy,")
print("# prologue and epilogue of the launch included.")
print(f"{'kernel':60s} {'mfma_busy':>11s} {'busy_cu':>11s} {'ratio':>7s} {'ratio/4':>8s}")
for k, (m, b) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if b > 0 and m > 0:
        print(f"{k[:60]:60s} {m:11.3e} {b:11.3e} {m / b:7.3f} {m / b / 4:8.3f}")
PY
rm -rf $O/pmc_busy
cat $O/r05_pmc_mfma_busy.txt
