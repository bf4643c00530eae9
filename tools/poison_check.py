"""Diagnostic: fill every `torch.empty*` allocation with NaN (floating point) / 0xFF (integers) and run training steps,
eager and from the captured hipGraph.  A kernel that reads memory nobody wrote turns the loss / the weights into NaN."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tinyedm_amd.ema import EMAOptimizer  # noqa: E402
from tinyedm_amd.graph import CapturedTrainStep  # noqa: E402
import tinyedm  # noqa: E402

_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        else:
            t.view(torch.uint8).fill_(0xFF) if t.is_contiguous() else None
    return t


def empty(*a, **k):
    return _poison(_empty(*a, **k))


def empty_like(*a, **k):
    return _poison(_empty_like(*a, **k))


dev = torch.device("cuda:0")
if "--prefill" in sys.argv:
    # every byte of free HBM becomes 0xFF (NaN as bf16 and as fp32) and goes back to the driver: fresh allocations
    # (and whatever lies beyond the end of a tensor) then hold NaN unless the driver scrubs
    free, total = torch.cuda.mem_get_info()
    blocks = []
    while True:
        free, _ = torch.cuda.mem_get_info()
        if free < (12 << 30):
            break
        blocks.append(torch.full((8 << 30,), 0xFF, dtype=torch.uint8, device=dev))
    print(f"prefilled {len(blocks) * 8} GiB with 0xFF", flush=True)
    torch.cuda.synchronize()
    del blocks
    torch.cuda.empty_cache()
    probe = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    print(f"fresh allocation after the prefill: {float((probe == 0xFF).float().mean()) * 100:.1f} % still 0xFF", flush=True)
    del probe
model, cfg = bench.build_model(dev, conditional="--cond" in sys.argv)
model.train()
base = model.configure_optimizers()["optimizer"]
opt = EMAOptimizer(base, device=dev, gamma=tinyedm.sigma_rel_to_gamma(model.ema_length), every_n_steps=model.every_n_steps)
g = torch.Generator().manual_seed(42)
batch = ((0.5 * torch.randn(128, 3, 32, 32, generator=g)).to(dev), torch.randint(0, 10, (128,), generator=g).to(dev))
if "--no-poison" not in sys.argv:
    torch.empty, torch.empty_like = empty, empty_like


def report(tag, loss):
    gn = float(base.arena.grad.norm())
    print(f"{tag}: loss {float(loss):.5f} |w| {float(base.arena.theta.norm()):.4f} |g| {gn:.4e}", flush=True)


for i in range(3):
    loss = model.training_step(batch, i)
    loss.backward()
    report(f"eager {i} (before Adam)", loss)
    opt.step()
    opt.zero_grad()
if "--graph" in sys.argv:
    cap = CapturedTrainStep(model, opt)
    for i in range(6):
        report(f"graph {i}", cap(batch))
