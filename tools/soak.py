"""Stability soak: N optimisation steps of the CIFAR-10 config on a fixed synthetic dataset of 4 batches (eager step with
the weight-gradient side stream, then the captured step), printing loss / weight norm every 50 steps.  Any NaN, or a loss
that does not fall on data this small, fails."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import tinyedm  # noqa: E402
from tinyedm_amd.ema import EMAOptimizer  # noqa: E402
from tinyedm_amd.graph import CapturedTrainStep  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev)
model.train()
oc = model.configure_optimizers()
base, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
g = torch.Generator().manual_seed(3)
# Gaussian images of std sigma_data are already denoised optimally by the skip path (loss floor 1.0): use a dataset the
# network can learn -- every image is one of four fixed patterns -- so that a falling loss means the whole step works
pats = 0.5 * torch.randn(4, 3, 32, 32, generator=g).sign()
data = [(pats[torch.randint(0, 4, (128,), generator=g)].to(dev), None) for _ in range(4)]
with torch.no_grad():
    model.denoiser.gain_out.fill_(0.5)      # EDM2 initialises the output gain to 0: nothing but that scalar learns at first
for p in base.param_groups:
    p["lr"] = 1e-2
hist = []
cap = None
for i in range(N):
    if i == N // 2:
        cap = CapturedTrainStep(model, opt)
    if cap is not None:
        loss = cap(data[i % 4])
    else:
        loss = model.training_step(data[i % 4], i)
        loss.backward()
        opt.step()
        opt.zero_grad()
    if i % 50 == 49 or i == N - 1:
        l, wn = float(loss), float(base.arena.theta.norm())
        hist.append(l)
        print(f"step {i + 1:4d} ({'graph' if cap is not None else 'eager'}): loss {l:.4f} |w| {wn:.2f}", flush=True)
        assert l == l and wn == wn, "non-finite"
assert hist[-1] < hist[0], ("loss did not fall", hist)
print("SOAK OK", hist[0], "->", hist[-1])
