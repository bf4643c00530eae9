#!/bin/bash
# kernel-trace A/B of the eager sampler (two solves) under an environment toggle, in ONE gpurun call:
#   tools/prof_sampler_ab.sh <dtype> <B> <tag_a> "<ENV=VAL ...>" <tag_b> "<ENV=VAL ...>"   -> gpurun_out/<tag>_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
DT=$1; B=$2
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
shift 2
while [ $# -ge 2 ]; do
  TAG=$1; ENVS=$2; shift 2
  rm -rf $O/prof_$TAG
  # (the toggles are exported into THIS shell: rocprofv3 must start python directly, never through `env`)
  for kv in $ENVS; do export "$kv"; done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/tools/sampler_profile.py $B $DT > $O/${TAG}.log 2>&1
  for kv in $ENVS; do unset "${kv%%=*}"; done
  cp $(ls $O/prof_$TAG/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
  rm -rf $O/prof_$TAG
  tail -1 $O/${TAG}.log
done
