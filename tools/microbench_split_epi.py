"""Output forms of ops.split_conv (round 6) on one 3x3 layer shape: us per launch for fp32 / fp32 + residual / pairs /
fp32 + pairs / fp32 + mp_silu pairs / the input halves of the next block's concatenated operands (dest).
    python tools/microbench_split_epi.py [B=512] [HW=32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = "cuda"
C = Cout = 256
xs = [torch.randn(B, HW, HW, 2 * C, device=dev).to(torch.bfloat16) for _ in range(2)]
rs = [torch.randn(B, HW, HW, Cout, device=dev) for _ in range(2)]
pk = ops.split_pack(torch.randn(Cout, C * 9, device=dev) / (C * 9) ** 0.5, 9)
cat = torch.empty(B, HW, HW, 4 * Cout, device=dev, dtype=torch.bfloat16)
sil = torch.empty_like(cat)
forms = {"fp32": dict(), "fp32 +R": dict(res=True), "pairs": dict(pairs_out=True), "pairs +R": dict(pairs_out=True, res=True),
         "fp32 + pairs +R": dict(also_pairs=True, res=True), "fp32 + silu pairs +R": dict(silu_pairs=True, res=True),
         "dest +R": dict(dest=(cat, sil), res=True)}
print(f"B={B} {HW}x{HW} {C}->{Cout} k9")
for name, kw in forms.items():
    kw = dict(kw)
    res = kw.pop("res", False)

    def run(i):
        return ops.split_conv(xs[i % 2], pk, 9, residual=rs[i % 2] if res else None, alpha=0.7, beta=0.7 if res else 0.0, **kw)
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(10):
        run(i)
    e.record()
    torch.cuda.synchronize()
    print(f"{name:24s} {s.elapsed_time(e) / 10 * 1e3:8.1f} us", flush=True)
