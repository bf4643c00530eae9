"""Does an HBM-streaming kernel (the fused Adam + EMA pass over the 35.6 M-parameter arena) run in the shadow of the MFMA-bound
3x3 convs when both are in flight on two streams?  Serial vs concurrent wall time of 14 conv launches + the optimizer pass cut
into NB arena slices (python tools/overlap_probe.py [NB])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
B = 128
x = torch.randn(B, 32, 32, 256, device=dev).to(torch.bfloat16)
wp = (torch.randn(9, 256, 256, device=dev) / 48).to(torch.bfloat16)
n = 35_600_000 // 64 * 64
theta, grad, m, v, ema = (torch.randn(n, device=dev) * 0.01 for _ in range(5))
side = torch.cuda.Stream()


def convs():
    for _ in range(14):
        ops.conv_igemm(x, wp, 9)


def adam():
    step = n // NB // 64 * 64
    for i in range(NB):
        lo, hi = i * step, (n if i == NB - 1 else (i + 1) * step)
        ops.adam_ema(theta[lo:hi], grad[lo:hi], m[lo:hi], v[lo:hi], ema[lo:hi], 1e-3, 0.9, 0.99, 1e-8, 3, 0.9)


def timed(fn, it=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        adam()
    convs()
    torch.cuda.current_stream().wait_stream(side)


tc, ta, tb = timed(convs), timed(adam), timed(both)
print(f"14 convs {tc:.3f} ms, Adam+EMA in {NB} slices {ta:.3f} ms, serial {tc + ta:.3f} ms, concurrent {tb:.3f} ms "
      f"(hidden: {(tc + ta - tb) / ta * 100:.0f} % of the optimizer pass)")
