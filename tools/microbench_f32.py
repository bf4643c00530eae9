"""fp32 evaluation kernels on the CIFAR-10 layer shapes: TFLOP/s against the 157.3 TF f32-MFMA peak.
    python tools/microbench_f32.py [--batch 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev, B = "cuda", a.batch


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters


for (HW, Cin, Cout, taps) in [(32, 256, 256, 9), (32, 512, 256, 9), (16, 256, 256, 9), (16, 512, 256, 9), (8, 256, 256, 9),
                              (32, 512, 256, 1), (16, 256, 768, 1), (16, 256, 256, 1), (8, 256, 768, 1)]:
    x = torch.randn(B, HW, HW, Cin, device=dev)
    w = torch.randn(Cout, Cin * taps, device=dev) / (Cin * taps) ** 0.5
    ms = timeit(lambda: ops.f32_conv(x, w, taps))
    fl = 2.0 * B * HW * HW * Cin * Cout * taps
    print(f"conv {HW:2d}x{HW:<2d} {Cin:3d}->{Cout:3d} k{taps}: {ms * 1e3:9.1f} us {fl / ms / 1e9:7.1f} TF/s ({fl / ms / 1e9 / 157.3:.2f} of peak)", flush=True)
for HW, C in ((16, 256), (8, 256), (16, 576), (8, 768)):
    qkv = torch.randn(B, HW, HW, 3 * C, device=dev)
    ms = timeit(lambda: ops.f32_attention(qkv, 4))
    fl = 4.0 * B * (HW * HW) ** 2 * C
    print(f"attention {HW}x{HW} d{C // 4}: {ms * 1e3:9.1f} us {fl / ms / 1e9:7.1f} TF/s", flush=True)
x = torch.randn(B, 32, 32, 256, device=dev)
print(f"pixelnorm_silu 32x32: {timeit(lambda: ops.f32_pixelnorm_silu(x)) * 1e3:.1f} us; silu {timeit(lambda: ops.f32_silu(x)) * 1e3:.1f} us", flush=True)
