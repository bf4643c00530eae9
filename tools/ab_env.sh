#!/bin/bash
# A/B of one environment switch inside ONE gpurun call (box-to-box spread is +-4 %: only same-box ratios mean anything):
#   tools/ab_env.sh EDM_FUSE_CAT 0 1 [bench flags]   -> gpurun_out/ab_<VAR>.txt
VAR=$1; A=$2; B=$3; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ab_$VAR.txt
: > $O
for rep in 1 2; do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 300 python3 $R/bench.py --steps 100 --warmup 10 --no-sampler --no-cpu-baseline --step-launch graph "$@" 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['value'])" >> $O || exit 1
  done
done
cat $O
