#!/bin/bash
# Steady-state rate of the drop-in training entry point against bench.py at the same batch, in ONE gpurun call:
#   tools/train_script_rate.sh  -> gpurun_out/train_script.txt  (copied to profiles/r05_train_script.txt)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/train_script.txt
: > $O
echo "# experiments/train.py --config-name=cifar10 (synthetic resident data, batch 256, 195 full batches + one of 80 per epoch)," >> $O
echo "# per-epoch rate of Trainer.fit (EDM_FIT_EPOCH_RATE=1), validation and callbacks off; then bench.py at the same batch" >> $O
for g in auto 1 0; do
  echo "## EDM_GRAPH=$g (auto = unset: fit() probes both forms and keeps the faster)" >> $O
  if [ $g = auto ]; then unset EDM_GRAPH; else export EDM_GRAPH=$g; fi
  EDM_FIT_EPOCH_RATE=1 timeout -k 10 400 python3 $R/experiments/train.py --config-name=cifar10 trainer.max_epochs=5 \
      trainer.check_val_every_n_epoch=1000 callbacks=null 2>&1 | grep "rate" >> $O || exit 1
done
echo "## bench.py --batch 256 (hipGraph replay of the same step, one resident batch)" >> $O
timeout -k 10 300 python3 $R/bench.py --batch 256 --steps 100 --warmup 10 --no-sampler --no-cpu-baseline 2>/dev/null \
  | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config'].get('step_launch'))" >> $O
cat $O
