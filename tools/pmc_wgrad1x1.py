"""HBM fetch of the 1x1 weight-gradient kernel per layer shape (run under `rocprofv3 --pmc FETCH_SIZE --output-format csv`):
one grouped launch per shape, every operand a fresh buffer (no Infinity-Cache hits from an earlier iteration), the
algorithmic bytes printed in dispatch order.  python tools/pmc_wgrad1x1.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

B = 128
shapes = [(16, 256, 768), (16, 256, 256), (16, 512, 256), (32, 512, 256), (8, 256, 768), (8, 512, 256)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for HW, Cin, Cout in shapes:
    x = torch.randn(B, HW, HW, Cin, device="cuda").to(torch.bfloat16)
    gy = torch.randn(B, HW, HW, Cout, device="cuda").to(torch.bfloat16)
    flush.zero_()                       # push the operands out of the L2 / Infinity Cache
    torch.cuda.synchronize()
    out = ops.conv_wgrad_1x1_group([(x, gy)])
    torch.cuda.synchronize()
    S = out[0].shape[0]
    print(f"{HW}x{HW} {Cin}->{Cout}: splits {S}, algorithmic read {2.0 * B * HW * HW * (Cin + Cout) / 1e6:.1f} MB, "
          f"slabs written {out[0].numel() * 4 / 1e6:.1f} MB", flush=True)

# the groups of a training step's backward pass (CIFAR-10 config: decoder 32x32 skips, then 16x16 and 8x8 attention /
# skip layers, then the encoder's attention layers), 16 layers per launch
layers = ([(32, 512, 256)] * 3 + [(16, 256, 256), (16, 256, 768), (16, 512, 256)] * 3 +
          [(8, 256, 256), (8, 256, 768), (8, 512, 256)] * 3 + [(8, 256, 256), (8, 256, 768)] +
          [(8, 256, 256), (8, 256, 768)] * 2 + [(16, 256, 256), (16, 256, 768)] * 2)
for g0 in range(0, len(layers), 16):
    grp = layers[g0:g0 + 16]
    pairs = []
    alg = 0.0
    for HW, Cin, Cout in grp:
        pairs.append((torch.randn(B, HW, HW, Cin, device="cuda").to(torch.bfloat16),
                      torch.randn(B, HW, HW, Cout, device="cuda").to(torch.bfloat16)))
        alg += 2.0 * B * HW * HW * (Cin + Cout)
    flush.zero_()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.conv_wgrad_1x1_group(pairs)
    e.record()
    torch.cuda.synchronize()
    print(f"group of {len(grp)} layers: algorithmic read {alg / 1e6:.1f} MB, {s.elapsed_time(e) * 1e3:.0f} us", flush=True)
