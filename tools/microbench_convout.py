import sys, torch
sys.path.insert(0,'/root/repo')
from tinyedm_amd import ops
dev='cuda'
B,HW,C,Co=128,32,256,3
x=torch.randn(B,HW,HW,C,device=dev).to(torch.bfloat16)
wh=torch.randn(Co,C,device=dev)/16
gain=torch.full((),0.7,device=dev); noisy=torch.randn(B,Co,HW,HW,device=dev); sigma=torch.rand(B,device=dev)+0.1
dD=torch.randn(B,Co,HW,HW,device=dev)
D,F=ops.conv_out_fwd(x,wh,gain,noisy,sigma,0.5)
def t(fn,it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
print("fwd", t(lambda: ops.conv_out_fwd(x,wh,gain,noisy,sigma,0.5)), "us; bwd (x + w)", t(lambda: ops.conv_out_bwd(x,wh,gain,F,dD,sigma,0.5)), "us")
gx,gwh,gg=ops.conv_out_bwd(x,wh,gain,F,dD,sigma,0.5)
print("checks", gx.float().abs().sum().item(), gwh.abs().sum().item(), gg.item())
