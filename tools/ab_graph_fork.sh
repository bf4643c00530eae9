#!/bin/bash
# captured step as one chain vs with the weight-gradient side stream as a branch of the graph, per config, ONE gpurun call
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${ROUND:-r06}_ab_graph_fork.txt
: > $O
BC="python3 $R/tools/bench_config.py"
for F in 0 1 0 1; do
  export EDM_GRAPH_FORK=$F
  echo "== EDM_GRAPH_FORK=$F" >> $O
  timeout -k 10 200 $BC cifar10 128 40 --graph >> $O 2>/dev/null
  timeout -k 10 200 $BC mnist 128 20 --graph >> $O 2>/dev/null
  timeout -k 10 200 $BC imagenet 176 8 --graph >> $O 2>/dev/null
  timeout -k 10 200 $BC imagenet 176 12 --graph --shape 4,32,32 >> $O 2>/dev/null
  timeout -k 10 200 $BC imagenet 704 8 --graph --shape 4,32,32 >> $O 2>/dev/null
done
unset EDM_GRAPH_FORK
echo "== eager loop" >> $O
timeout -k 10 200 $BC imagenet 176 8 >> $O 2>/dev/null
timeout -k 10 200 $BC imagenet 176 12 --shape 4,32,32 >> $O 2>/dev/null
cat $O
