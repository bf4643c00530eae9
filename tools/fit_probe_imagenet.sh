#!/bin/bash
# one fit() of the imagenet config showing which launch form Trainer.fit's probe settles on there (the 272 M net on
# 4 x 64 x 64 latents; accumulate_grad_batches=1 so that the captured step is eligible; 50 batches of 176 per epoch):
#   tools/fit_probe_imagenet.sh -> gpurun_out/r06_fit_probe_imagenet.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${ROUND:-r06}_fit_probe_imagenet.txt
: > $O
for g in auto 1 0; do
  echo "## EDM_GRAPH=$g (auto = unset: fit() probes both forms, 2 + 8 eager and 1 + 8 replayed steps, and keeps the faster)" >> $O
  if [ $g = auto ]; then unset EDM_GRAPH; else export EDM_GRAPH=$g; fi
  EDM_FIT_EPOCH_RATE=1 timeout -k 10 400 python3 $R/experiments/train.py --config-name=imagenet trainer.max_epochs=3 \
      trainer.accumulate_grad_batches=1 trainer.check_val_every_n_epoch=1000 callbacks=null datamodule.num_samples=8800 2>&1 \
      | grep "rate" >> $O || exit 1
done
cat $O
