"""In-kernel shader clock of the 3x3 igemm v3 under sustained load (s_memtime / s_memrealtime)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops, _lib
B, HW, Cin, Cout = 128, 32, 256, 256
x = torch.randn(B, HW, HW, Cin, device="cuda").to(torch.bfloat16)
wp = (torch.randn(9, Cout, Cin, device="cuda") / 48).to(torch.bfloat16)
ops.IGEMM_VERSION = 3
y = ops.conv_igemm(x, wp, 9)
P = ctypes.c_void_p
st = P(torch.cuda.current_stream().cuda_stream)
dbg = torch.zeros(4, device="cuda", dtype=torch.int64)
t0 = time.time()
while time.time() - t0 < 2.0:          # >= 2 s of back-to-back launches on random data
    for _ in range(50):
        ops.conv_igemm(x, wp, 9)
    torch.cuda.synchronize()
dbg.zero_()
for _ in range(20):
    _lib.call("edm_conv_igemm_v3_clock", P(x.data_ptr()), P(wp.data_ptr()), P(y.data_ptr()), B, HW, HW, Cin, Cout, P(dbg.data_ptr()), st)
torch.cuda.synchronize()
c, r, n = dbg.cpu().tolist()[:3]
print(f"in-kernel clock {c / r * 0.1:.3f} GHz; per-workgroup kernel time {r / n / 100:.1f} us ({n} workgroups)")
print(f"=> dense bf16 MFMA peak at this clock: {2500 * (c / r * 0.1) / 2.4:.0f} TFLOP/s")
