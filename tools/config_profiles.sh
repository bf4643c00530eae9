#!/bin/bash
# BASELINE.json configs[0], [2], [3], [4] beside the headline config: throughput with whole-step MFMA / HBM fractions, the
# per-kernel breakdown of a training step (rocprofv3 kernel trace of the replayed graph), and the 32-step sampler of the
# 272 M net -- ONE gpurun call:   ROUND=r06 tools/config_profiles.sh   -> gpurun_out/r06_config_*; copy into profiles/
set -e
RD=${ROUND:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=$R/gpurun_out
mkdir -p $O
T=$O/${RD}_config_throughput.txt
: > $T
BC="python3 $R/tools/bench_config.py"
run() { timeout -k 10 280 "$@" >> $T 2>> $O/${RD}_config.err; }
# ---- throughput (eager loop and replayed graph) + sampler legs
run $BC mnist 128 20 --fwd-gflop 20.1
run $BC mnist 128 20 --graph --fwd-gflop 20.1
run $BC cifar10_cond 256 20 --graph --fwd-gflop 27.0 --fwd-mb 31.7
# ImageNet-64, pixel space (BASELINE configs[3]): the default Denoiser on 3 x 64 x 64
run $BC imagenet 176 8 --shape 3,64,64 --fwd-gflop 192.9 --fwd-mb 168.9
run $BC imagenet 176 8 --shape 3,64,64 --graph --fwd-gflop 192.9 --fwd-mb 168.9 --sampler bf16 176
# the YAML's 4 x 64 x 64 latents
run $BC imagenet 176 8 --fwd-gflop 192.9 --fwd-mb 168.9
run $BC imagenet 176 8 --graph --fwd-gflop 192.9 --fwd-mb 168.9
# BASELINE configs[4]: 32 x 32 x 4 latents (ImageNet-256 through the SD-VAE), training + the 32-step sampler of the 272 M net
run $BC imagenet 176 8 --shape 4,32,32 --fwd-gflop 48.0 --fwd-mb 42.3
run $BC imagenet 176 8 --shape 4,32,32 --graph --fwd-gflop 48.0 --fwd-mb 42.3
run $BC imagenet 704 8 --shape 4,32,32 --fwd-gflop 48.0 --fwd-mb 42.3
run $BC imagenet 704 8 --shape 4,32,32 --graph --fwd-gflop 48.0 --fwd-mb 42.3 --sampler bf16 512
run $BC imagenet 8 1 --no-train --shape 4,32,32 --fwd-gflop 48.0 --sampler f32x3 512
cat $T
# ---- kernel breakdown of one replayed training step per config
for spec in "mnist 128 mnist ''" "imagenet 176 latent64 ''" "imagenet 704 latent32 --shape 4,32,32" "imagenet 176 pixel64 --shape 3,64,64"; do
  set -- $spec
  NAME=$1; B=$2; TAG=$3; shift 3
  EXTRA="$@"; [ "$EXTRA" = "''" ] && EXTRA=""
  rm -rf $O/prof_cfg
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python3 $R/tools/bench_config.py $NAME $B 6 --graph $EXTRA > $O/${RD}_config_${TAG}_prof.log 2>&1
  python3 $R/tools/step_breakdown.py $(ls $O/prof_cfg/*/*kernel_trace.csv | head -1) 40 > $O/${RD}_config_${TAG}_step_breakdown.txt
  rm -rf $O/prof_cfg
  head -3 $O/${RD}_config_${TAG}_step_breakdown.txt
done
# ---- kernel stats of the 272 M net's sampler on 32 x 32 x 4 (eager loop, one solve: 63 evaluations)
rm -rf $O/prof_cfg
cat > /tmp/sampler_cfg.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["EDM_ROOT"])
import tinyedm
from tinyedm.config import compose, instantiate
cfg = compose("imagenet", os.path.join(os.environ["EDM_ROOT"], "experiments", "conf"))
cfg.model.denoiser.in_channels = cfg.model.denoiser.out_channels = 4
dev = torch.device("cuda:0")
model = instantiate(cfg.model).to(dev).eval()
model.denoiser.set_eval_dtype(sys.argv[1])
B = int(sys.argv[2])
x0 = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(7)).to(dev)
lab = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(8)).to(dev)
solver = tinyedm.DeterministicSolver(num_steps=32)
solver.solve(model, x0, lab)
torch.cuda.synchronize()
PY
for DT in bf16 f32x3; do
  rm -rf $O/prof_cfg
  EDM_ROOT=$R timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python3 /tmp/sampler_cfg.py $DT 512 > $O/${RD}_config_latent32_sampler_${DT}.log 2>&1
  cp $(ls $O/prof_cfg/*/*kernel_stats.csv | head -1) $O/${RD}_config_latent32_sampler_${DT}_kernel_stats.csv
  rm -rf $O/prof_cfg
done
echo done
