#!/bin/bash
# EDM_MOD_DEFER_UNFUSED (unfused modulation backward: raw gradient into the shared buffer, no finish launch per block) on MNIST,
# ONE gpurun call -> gpurun_out/ab_mod_defer.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ab_mod_defer.txt
: > $O
for rep in 1 2; do
  for v in 0 1; do
    echo "== EDM_MOD_DEFER_UNFUSED=$v" >> $O
    env EDM_MOD_DEFER_UNFUSED=$v timeout -k 10 200 python3 $R/tools/bench_config.py mnist 128 30 --graph --fwd-gflop 20.1 >> $O 2>/dev/null || exit 1
  done
done
cat $O
