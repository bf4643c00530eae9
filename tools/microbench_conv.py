"""Micro-benchmark of the MFMA kernels on the CIFAR-10 layer shapes (B=128): TFLOP/s per kernel/shape.
Usage: python tools/microbench_conv.py [--iters 20] [--only igemm|wgrad|all]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="all")
ap.add_argument("--batch", type=int, default=128)
a = ap.parse_args()
dev = "cuda"
B = a.batch
shapes = [(32, 256, 256, 9), (32, 512, 256, 9), (16, 256, 256, 9), (16, 512, 256, 9), (8, 256, 256, 9), (8, 512, 256, 9),
          (32, 512, 256, 1), (16, 256, 768, 1), (16, 256, 256, 1), (8, 256, 768, 1)]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters


for (HW, Cin, Cout, taps) in shapes:
    x = torch.randn(B, HW, HW, Cin, device=dev).to(torch.bfloat16)
    gy = torch.randn(B, HW, HW, Cout, device=dev).to(torch.bfloat16)
    wp = (torch.randn(taps, Cout, Cin, device=dev) / (Cin * taps) ** 0.5).to(torch.bfloat16)
    fl = 2.0 * B * HW * HW * Cin * Cout * taps
    line = f"{HW:2d}x{HW:<2d} {Cin:3d}->{Cout:3d} k{taps}: "
    if a.only in ("all", "igemm"):
        ms = timeit(lambda: ops.conv_igemm(x, wp, taps))
        line += f"igemm {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s   "
    if a.only in ("all", "wgrad"):
        ms = timeit(lambda: ops.conv_wgrad(x, gy, taps))
        line += f"wgrad {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s"
    if a.only in ("all", "wgrad3") and taps == 9:
        w = torch.randn(Cout, Cin, 3, 3, device=dev)
        gr = torch.zeros_like(w)
        for n in (1, 4):                 # one layer per launch / four layers of this shape per grouped launch
            items = [(x, gy, w, gr, None, 1.0, True)] * n
            ms = timeit(lambda: ops.wgrad3_group(items))
            line += f"wgrad3 x{n} {ms * 1e3 / n:8.1f} us/layer {n * fl / ms / 1e9:7.1f} TF/s   "
    print(line, flush=True)
