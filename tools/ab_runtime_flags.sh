R=${GRAFT_REPO_ROOT:-$PWD}
run() { timeout -k 10 300 python3 $R/bench.py --steps 100 --warmup 10 --no-sampler --no-cpu-baseline --step-launch graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  run "unset"
  HIP_FORCE_DEV_KERNARG=1 run "HIP_FORCE_DEV_KERNARG=1"
  HIP_FORCE_DEV_KERNARG=0 run "HIP_FORCE_DEV_KERNARG=0"
  DEBUG_HIP_KERNARG_COPY_OPT=0 run "DEBUG_HIP_KERNARG_COPY_OPT=0"
  DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 run "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1"
  AMD_OPT_FLUSH=0 run "AMD_OPT_FLUSH=0"
  DEBUG_HIP_GRAPH_BATCH_SIZE=1024 run "DEBUG_HIP_GRAPH_BATCH_SIZE=1024"
  DEBUG_CLR_MAX_BATCH_SIZE=2048 run "DEBUG_CLR_MAX_BATCH_SIZE=2048"
done
