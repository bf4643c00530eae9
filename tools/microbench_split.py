"""Micro-benchmark of ops.split_conv (the "f32x3" evaluation's convolution) on the CIFAR-10 layer shapes: us and TFLOP/s
(counting the three bf16 passes) per shape, rotating operand sets so that the 256-MB Infinity Cache does not serve re-reads.
A/B of the round-6 dispatch (k_conv3x3_s for the 8x8-class layers, k_conv_igemm2 for the 1x1 layers) in ONE call:
    EDM_SPLIT_FAST=0 python tools/microbench_split.py 256 && python tools/microbench_split.py 256"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda"
shapes = [(8, 256, 256, 9), (8, 512, 256, 9), (16, 256, 256, 9), (32, 256, 256, 9),
          (32, 512, 256, 1), (16, 512, 256, 1), (8, 512, 256, 1), (16, 256, 768, 1), (8, 256, 768, 1), (16, 256, 256, 1),
          (8, 256, 256, 1)]
print(f"B={B} EDM_SPLIT_FAST={os.environ.get('EDM_SPLIT_FAST', '1')}")
for HW, Cin, Cout, taps in shapes:
    nset = max(2, min(8, int(600e6 // (B * HW * HW * (2 * Cin * 2 + Cout * 4)))))
    xs = [torch.randn(B, HW, HW, 2 * Cin, device=dev).to(torch.bfloat16) for _ in range(nset)]
    rs = [torch.randn(B, HW, HW, Cout, device=dev) for _ in range(nset)]
    pk = ops.split_pack(torch.randn(Cout, Cin * taps, device=dev) / (Cin * taps) ** 0.5, taps)
    for res in (False, True):
        def run(i):
            return ops.split_conv(xs[i % nset], pk, taps, residual=rs[i % nset] if res else None, alpha=0.7,
                                  beta=0.7 if res else 0.0)
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        iters = 20
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters):
            run(i)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        fl = 3 * 2.0 * B * HW * HW * Cin * Cout * taps
        print(f"{HW:2d}x{HW:<2d} {Cin:3d}->{Cout:3d} k{taps}{' +R' if res else '   '}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s",
              flush=True)
