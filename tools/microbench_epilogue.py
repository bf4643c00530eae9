"""What the fused epilogues of the 3x3 implicit-GEMM kernels cost: the same conv timed plain, with the forward
modulation (conv3x3_mod), with the modulation backward (conv3x3_modbwd) and with the mp_silu backward
(conv3x3_silubwd), on the CIFAR-10 layer shapes at B=128.
Usage: python tools/microbench_epilogue.py [--iters 20] [--batch 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=128)
a = ap.parse_args()
dev = "cuda"
B = a.batch


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
    return s.elapsed_time(e) / a.iters * 1e3


for (HW, Cin, Cout) in [(32, 256, 256), (32, 512, 256), (32, 256, 512), (16, 256, 256), (16, 512, 256), (16, 256, 512),
                        (8, 256, 256), (8, 512, 256), (8, 256, 512)]:
    x = torch.randn(B, HW, HW, Cin, device=dev).to(torch.bfloat16)
    other = torch.randn(B, HW, HW, Cout, device=dev).to(torch.bfloat16)
    extra = torch.randn(B, HW, HW, Cout, device=dev).to(torch.bfloat16)
    wp = (torch.randn(9, Cout, Cin, device=dev) / (Cin * 9) ** 0.5).to(torch.bfloat16)
    lin = torch.randn(B, Cout, device=dev)
    gain = torch.full((), 0.3, device=dev)
    gm = torch.zeros(B, Cout, device=dev)
    fl = 2.0 * B * HW * HW * Cin * Cout * 9
    marked = ops.conv3x3_mod(x, wp, lin, gain, 0.13, 1, 2, 3, want_u=True, mark_dropped=True)[0]
    t = {
        "plain": timeit(lambda: ops.conv_igemm(x, wp, 9)),
        "plain+res": timeit(lambda: ops.conv_igemm(x, wp, 9, residual=other, alpha=0.7, beta=0.7)),
        "mod(u,a2)": timeit(lambda: ops.conv3x3_mod(x, wp, lin, gain, 0.13, 1, 2, 3, want_u=True)),
        "mod(a2)": timeit(lambda: ops.conv3x3_mod(x, wp, lin, gain, 0.13, 1, 2, 3, want_u=False)),
        "modbwd": timeit(lambda: ops.conv3x3_modbwd(x, wp, 1.0, other, lin, gain, 0.13, 1, 2, 3, gm_out=gm)),
        "modbwd marked": timeit(lambda: ops.conv3x3_modbwd(x, wp, 1.0, marked, lin, gain, 0.13, 1, 2, 3, gm_out=gm, u_marked=True)),
        "mod(u,a2) marks": timeit(lambda: ops.conv3x3_mod(x, wp, lin, gain, 0.13, 1, 2, 3, want_u=True, mark_dropped=True)),
        "mod(u,a2) p=0": timeit(lambda: ops.conv3x3_mod(x, wp, lin, gain, 0.0, 1, 2, 3, want_u=True)),
        "modbwd p=0": timeit(lambda: ops.conv3x3_modbwd(x, wp, 1.0, other, lin, gain, 0.0, 1, 2, 3, gm_out=gm)),
        "silubwd": timeit(lambda: ops.conv3x3_silubwd(x, wp, other)),
        "silubwd+add": timeit(lambda: ops.conv3x3_silubwd(x, wp, other, extra, 0.5)),
    }
    print(f"{HW:2d}x{HW:<2d} {Cin:3d}->{Cout:3d} ({fl / 1e9:6.1f} GF): " + "  ".join(f"{k} {v:6.1f}" for k, v in t.items()) + "  us",
          flush=True)
