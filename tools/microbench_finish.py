"""Time edm_wgrad_finish on the layer shapes of the CIFAR-10 net (slab bytes / time = effective HBM rate)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402


def run(S, O, I, taps, iters=50):
    dev = "cuda"
    slabs = torch.randn(S, taps, O, I, device=dev)
    w = torch.randn(O, I * taps, device=dev)
    out = torch.zeros_like(w)
    for _ in range(3):
        ops.wgrad_finish(slabs, w, taps, I, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.wgrad_finish(slabs, w, taps, I, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    mb = slabs.numel() * 4 / 1e6
    print(f"S={S:4d} O={O:4d} I={I:4d} taps={taps}: {us:7.1f} us  {mb:7.1f} MB  {mb / us * 1e-3 * 1e3:6.2f} GB/ms = {mb / us:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    run(16, 256, 256, 9)
    run(8, 256, 512, 9)
    run(64, 256, 256, 1)
    run(43, 768, 256, 1)
    run(64, 256, 512, 1)
