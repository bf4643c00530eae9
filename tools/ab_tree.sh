#!/bin/bash
# A/B of the training step between this tree and a second checkout (e.g. `git archive HEAD | tar -x -C ab/old` plus its
# built library) inside ONE gpurun call: tools/ab_tree.sh ab/old [rounds].  Alternates new, old, new, old ...
OLD="$1"; R="${2:-2}"
run() { (cd "$1" && python bench.py --steps 40 --warmup 8 --no-sampler --no-cpu-baseline --step-launch graph 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for i in $(seq 1 $R); do
  echo "new $(run .) ms/step"
  echo "old $(run $OLD) ms/step"
done
