#!/bin/bash
# EDM_W3_GROUP (3x3 layers per grouped weight-gradient launch; the table holds 48) on the headline step, ONE gpurun call
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ab_w3_group.txt
: > $O
for rep in 1 2; do
  for v in 16 32 48; do
    env EDM_W3_GROUP=$v timeout -k 10 300 python3 $R/bench.py --steps 100 --warmup 10 --no-sampler --no-cpu-baseline --step-launch graph 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('EDM_W3_GROUP=$v', d['ms_per_step'], d['value'], d['config'].get('final_loss'))" >> $O || exit 1
  done
done
cat $O
