#!/bin/bash
# EDM_FIN_PREFETCH (k_wgrad_finish_multi: master row + old gradient loaded while the slab loads are in flight; =0: late loads):
# the launches' duration from a kernel trace of the replayed step, both settings in ONE gpurun call -> gpurun_out/ab_fin_prefetch.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
: > $O/ab_fin_prefetch.txt
for v in 0 1 0 1; do
  rm -rf $O/prof_fin
  export EDM_FIN_PREFETCH=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fin -- python3 $R/bench.py --steps 20 --warmup 5 --no-sampler --no-cpu-baseline --step-launch graph > $O/ab_fin_$v.json 2> $O/ab_fin_$v.log || exit 1
  python3 $R/tools/step_breakdown.py $(ls $O/prof_fin/*/*kernel_trace.csv | head -1) 60 > $O/ab_fin_bd.txt
  echo "EDM_FIN_PREFETCH=$v  $(head -1 $O/ab_fin_bd.txt)  |  $(grep k_wgrad_finish_multi $O/ab_fin_bd.txt)" >> $O/ab_fin_prefetch.txt
done
rm -rf $O/prof_fin
cat $O/ab_fin_prefetch.txt
