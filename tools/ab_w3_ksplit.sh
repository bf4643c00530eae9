#!/bin/bash
# EDM_W3_KSPLIT (k_wgrad3: K shares for layers whose tiles do not fill a team of eight; =0: off) on the configurations that have
# such layers, inside ONE gpurun call:   tools/ab_w3_ksplit.sh   -> gpurun_out/ab_w3_ksplit.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ab_w3_ksplit.txt
echo "# EDM_W3_KSPLIT (k_wgrad3: K shares per tile where a layer's tiles leave team members idle; =0: off), one gpurun call, tools/bench_config.py" > $O
BC="python3 $R/tools/bench_config.py"
for rep in 1 2; do
  for v in 0 8; do
    echo "== EDM_W3_KSPLIT=$v" >> $O
    env EDM_W3_KSPLIT=$v timeout -k 10 200 $BC mnist 128 30 --graph --fwd-gflop 20.1 >> $O 2>/dev/null || exit 1
    env EDM_W3_KSPLIT=$v timeout -k 10 280 $BC imagenet 176 8 --graph --fwd-gflop 192.9 >> $O 2>/dev/null || exit 1
    echo "ksplit=$v rep=$rep done"
  done
done
for v in 0 8; do
  echo "== EDM_W3_KSPLIT=$v" >> $O
  env EDM_W3_KSPLIT=$v timeout -k 10 280 $BC imagenet 704 8 --shape 4,32,32 --graph --fwd-gflop 48.0 >> $O 2>/dev/null || exit 1
  env EDM_W3_KSPLIT=$v timeout -k 10 280 $BC mnist 128 30 --fwd-gflop 20.1 >> $O 2>/dev/null || exit 1
  echo "ksplit=$v tail done"
done
cat $O
