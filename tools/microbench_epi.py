"""Cost of the fused epilogues of the 3x3 kernel at B=128, 32x32, 256->256: plain conv, conv + residual, forward with the
modulation/SiLU/dropout epilogue (pdrop 0 and 0.13, with / without the pre-activation output), modulation-backward and
SiLU-backward dgrads."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

dev = "cuda"
B, HW, C = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B, HW, HW, C, generator=g).to(torch.bfloat16).to(dev)
r = torch.randn(B, HW, HW, C, generator=g).to(torch.bfloat16).to(dev)
wp = (torch.randn(9, C, C, generator=g) / 48).to(torch.bfloat16).to(dev)
lin = torch.randn(B, C, generator=g).to(dev)
gain = torch.tensor(0.3, device=dev)
fl = 2.0 * B * HW * HW * C * C * 9


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


cases = {
    "plain": lambda: ops.conv_igemm(x, wp, 9),
    "plain + residual": lambda: ops.conv_igemm(x, wp, 9, residual=r, alpha=0.7, beta=0.3),
    "fwd mod p=0.13 (u and a2)": lambda: ops.conv3x3_mod(x, wp, lin, gain, 0.13, 1, 2, 3, want_u=True),
    "fwd mod p=0 (u and a2)": lambda: ops.conv3x3_mod(x, wp, lin, gain, 0.0, 1, 2, 3, want_u=True),
    "fwd mod p=0 (a2 only, eval)": lambda: ops.conv3x3_mod(x, wp, lin, gain, 0.0, 1, 2, 3, want_u=False),
    "dgrad + mod bwd p=0.13": lambda: ops.conv3x3_modbwd(x, wp, 0.8, r, lin, gain, 0.13, 1, 2, 3),
    "dgrad + silu bwd (+add)": lambda: ops.conv3x3_silubwd(x, wp, r, r, 0.5),
    "UNFUSED fwd: conv, then mod/silu/drop": lambda: ops.mod_silu_drop_fwd(ops.conv_igemm(x, wp, 9), lin, gain, 0.13, 1, 2, 3),
    "UNFUSED bwd: dgrad, then mod bwd": lambda: ops.mod_silu_drop_bwd(r, lin, gain, ops.conv_igemm(x, wp, 9, alpha=0.8), 0.13, 1, 2, 3),
    "UNFUSED bwd: dgrad, then silu bwd": lambda: ops.silu_bwd(r, ops.conv_igemm(x, wp, 9), r, 0.5),
}
for name, fn in cases.items():
    ms = timeit(fn)
    print(f"{name:40s} {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)
