"""Timing-only ablations of igemm v2 (3x3, 32x32, 256->256, B=128): which phase dominates?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops, _lib
B, HW, Cin, Cout = 128, 32, 256, 256
x = torch.randn(B, HW, HW, Cin, device="cuda").to(torch.bfloat16)
wp = (torch.randn(9, Cout, Cin, device="cuda") / 48).to(torch.bfloat16)
ops.IGEMM_VERSION = 2
y = ops.conv_igemm(x, wp, 9)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = ctypes.c_void_p
names = {0: "full", 1: "no MFMA", 2: "no DMA", 4: "no frag reads/MFMA (DMA+barrier only)", 6: "no DMA, no reads (loop+barrier)",
         8: "no barrier", 10: "no DMA, no barrier (reads+MFMA only)"}
res = {m: [] for m in names}
for rnd in range(5):
    for m in names:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            _lib.call("edm_conv_igemm_v2_ablate", P(x.data_ptr()), P(wp.data_ptr()), P(y.data_ptr()), B, HW, HW, Cin, Cout, m, st)
        e.record()
        torch.cuda.synchronize()
        res[m].append(s.elapsed_time(e) / 5 * 1e3)
for m, n in names.items():
    v = sorted(res[m])
    print(f"mode {m:2d} {n:42s} median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f} us")
