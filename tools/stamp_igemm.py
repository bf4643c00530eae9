"""Where does an igemm-v2 loop iteration spend its cycles?  (diagnostic build with s_memtime stamps)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops, _lib
B, HW, Cin, Cout = 128, 32, 256, 256
x = torch.randn(B, HW, HW, Cin, device="cuda").to(torch.bfloat16)
wp = (torch.randn(9, Cout, Cin, device="cuda") / 48).to(torch.bfloat16)
os.environ["EDM_IGEMM"] = "2"
ops.IGEMM_VERSION = 2
y = ops.conv_igemm(x, wp, 9)
dbg = torch.zeros(8, device="cuda", dtype=torch.int64)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    dbg.zero_()
    _lib.call("edm_conv_igemm_v2_stamp", ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
              ctypes.c_void_p(y.data_ptr()), B, HW, HW, Cin, Cout, ctypes.c_void_p(dbg.data_ptr()), st)
    torch.cuda.synchronize()
d = dbg.cpu().tolist()
waves, iters = d[5], (Cin // 32) * 9
names = ["vmcnt wait", "barrier", "dma issue", "frag reads", "mfma issue"]
tot = sum(d[:5])
for n, v in zip(names, d[:5]):
    print(f"{n:12s} {v / waves / iters:8.1f} cycles/iter/wave  {100.0 * v / tot:5.1f} %")
print(f"total {tot / waves / iters:.1f} cycles per iteration per wave ({waves} waves, {iters} iterations)")
