"""Time the attention kernels on the CIFAR-10 shapes: python tools/microbench_attn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402


def run(B, H, W, heads, hd=64, iters=30):
    C = heads * hd
    qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
    gy = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    y = ops.attention_fwd(qkv, heads)
    ops.attention_bwd(qkv, y, gy, heads)
    torch.cuda.synchronize()
    res = []
    for fn in (lambda: ops.attention_fwd(qkv, heads), lambda: ops.attention_bwd(qkv, y, gy, heads)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / iters)
    N = H * W
    gf = 4.0 * B * heads * N * N * hd / 1e9
    print(f"B={B} {H}x{W} heads={heads} d={hd}: fwd {res[0]:7.1f} us ({gf / res[0] * 1e3:6.1f} TF/s)  bwd {res[1]:7.1f} us "
          f"({2.5 * gf / res[1] * 1e3:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    run(128, 16, 16, 4)
    run(128, 8, 8, 4)
    run(256, 16, 16, 4)
