"""Which torch (aten) kernels are still launched inside a training step, and from where: one eager step under torch.profiler
with Python stacks; prints every aten op that launched a device kernel with the innermost frames of tinyedm_amd / bench code.
    python tools/stray_ops.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tinyedm_amd.ddp import GradReducer  # noqa: E402
from tinyedm_amd.ema import EMAOptimizer  # noqa: E402
import tinyedm  # noqa: E402

dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev)
model.train()
base = model.configure_optimizers()["optimizer"]
opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps) if model.use_ema else base
red = GradReducer(base.arena)
x = 0.5 * torch.randn(128, 3, 32, 32, device=dev)


def step(i):
    loss = model.training_step((x, None), i)
    model.backward(loss)
    base.grad_scale = red.finish()
    opt.step()
    opt.zero_grad()


opt.zero_grad()
for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(4)
    torch.cuda.synchronize()
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if not ev.key.startswith("aten::") or t <= 0:
        continue
    frames = [f for f in (ev.stack or []) if "tinyedm" in f or "bench" in f or "tools/" in f][:3]
    print(f"{ev.key:28s} x{ev.count:3d} {t:8.1f} us  {ev.input_shapes if ev.input_shapes else ''}")
    for f in frames:
        print("      ", f)
