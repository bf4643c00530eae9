import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops
B=128
x = torch.randn(B,32,32,32, device="cuda").to(torch.bfloat16)
wp = (torch.randn(9,256,32, device="cuda")/17).to(torch.bfloat16)
def timeit(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e3
for v in (0,1,3):
    ops.IGEMM_VERSION=v
    print(v, ops._igemm_entry(B*1024, 32, 256, 9, 32), f"{timeit(lambda: ops.conv_igemm(x, wp, 9)):.1f} us")
ops.IGEMM_VERSION=0
x64 = torch.randn(B,32,32,64, device="cuda").to(torch.bfloat16)
wp64 = (torch.randn(9,256,64, device="cuda")/24).to(torch.bfloat16)
print("Cin=64 auto", ops._igemm_entry(B*1024, 32, 256, 9, 64), f"{timeit(lambda: ops.conv_igemm(x64, wp64, 9)):.1f} us")
