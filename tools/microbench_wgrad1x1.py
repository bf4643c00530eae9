"""Grouped 1x1 weight gradient (k_wgrad1x1_group) against the tile count of a layer: us and algorithmic TB/s (every operand
byte once) for one layer shape per launch, rotating operand sets (no Infinity-Cache re-reads), and for the three groups of a
CIFAR-10 training step.  Is the kernel bound by HBM or by the re-reads of dY / X its 256x128 tiles make through L2?
    python tools/microbench_wgrad1x1.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"


def timeit(fn, iters=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def operands(HW, Cin, Cout):
    return (torch.randn(B, HW, HW, Cin, device=dev).to(torch.bfloat16), torch.randn(B, HW, HW, Cout, device=dev).to(torch.bfloat16))


for HW, Cin, Cout, nl in [(32, 128, 256, 3), (32, 256, 256, 3), (32, 512, 256, 3), (32, 1024, 256, 2), (32, 256, 128, 3), (32, 256, 768, 2),
                          (16, 256, 768, 8), (16, 256, 256, 8), (16, 512, 256, 8), (8, 256, 768, 16), (8, 512, 256, 16)]:
    nset = 3
    sets = [[operands(HW, Cin, Cout) for _ in range(nl)] for _ in range(nset)]
    ms = timeit(lambda i: ops.conv_wgrad_1x1_group(sets[i % nset]))
    alg = nl * 2.0 * B * HW * HW * (Cin + Cout)
    fl = nl * 2.0 * B * HW * HW * Cin * Cout
    tiles = ((Cout + 255) // 256) * ((Cin + 127) // 128)
    print(f"{nl:2d} x ({HW:2d}x{HW:<2d} {Cin:4d}->{Cout:3d}), {tiles} tiles per split: {ms * 1e3:7.1f} us  {alg / ms / 1e9:6.2f} TB/s algorithmic  "
          f"{fl / ms / 1e9:6.1f} TF/s", flush=True)
    del sets

layers = ([(32, 512, 256)] * 3 + [(16, 256, 256), (16, 256, 768), (16, 512, 256)] * 3 +
          [(8, 256, 256), (8, 256, 768), (8, 512, 256)] * 3 + [(8, 256, 256), (8, 256, 768)] +
          [(8, 256, 256), (8, 256, 768)] * 2 + [(16, 256, 256), (16, 256, 768)] * 2)
tot = 0.0
for g0 in range(0, len(layers), 16):
    grp = layers[g0:g0 + 16]
    sets = [[operands(*l) for l in grp] for _ in range(2)]
    ms = timeit(lambda i: ops.conv_wgrad_1x1_group(sets[i % 2]))
    alg = sum(2.0 * B * HW * HW * (Cin + Cout) for HW, Cin, Cout in grp)
    tot += ms
    print(f"step group of {len(grp)} layers: {ms * 1e3:7.1f} us  {alg / ms / 1e9:6.2f} TB/s algorithmic", flush=True)
    del sets
print(f"step total {tot * 1e3:.1f} us")
