"""Run ONE kernel shape repeatedly (for rocprofv3 --pmc passes): python tools/microbench_one.py igemm|wgrad3|s HW Cin Cout [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

kind, HW, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
B = 128
x = torch.randn(B, HW, HW, Cin, device="cuda").to(torch.bfloat16)
gy = torch.randn(B, HW, HW, Cout, device="cuda").to(torch.bfloat16)
wp = (torch.randn(9, Cout, Cin, device="cuda") / (Cin * 9) ** 0.5).to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda")
gr = torch.zeros_like(w)
def run():
    if kind == "igemm":
        ops.conv_igemm(x, wp, 9)
    else:
        ops.wgrad3_group([(x, gy, w, gr, None, 1.0, True)] * 4)


run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    run()
e.record()
torch.cuda.synchronize()
n = 1 if kind == "igemm" else 4
fl = 2.0 * B * HW * HW * Cin * Cout * 9 * n
ms = s.elapsed_time(e) / iters
print(f"{kind} {HW}x{HW} {Cin}->{Cout}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.0f} TF/s  (ablate={os.environ.get('EDM_W3_ABLATE', os.environ.get('EDM_V4_ABLATE', '0'))})", flush=True)
