#!/bin/bash
# A/B of the training step inside ONE gpurun call (box-to-box spread is +-4 %): tools/ab_bench.sh "ENV_A" "ENV_B" [rounds]
# each leg: bench.py --no-sampler --no-cpu-baseline --step-launch graph, 40 timed steps; alternates A, B, A, B ...
A="$1"; B="$2"; R="${3:-2}"
for i in $(seq 1 $R); do
  for leg in A B; do
    if [ $leg = A ]; then E="$A"; else E="$B"; fi
    ms=$(env $E python bench.py --steps 40 --warmup 8 --no-sampler --no-cpu-baseline --step-launch graph 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$leg [$E] $ms ms/step"
  done
done
