"""Minimal probe: can an RCCL all-reduce be captured into a hipGraph here?  One rank; variants chosen by argv[1]:
  same   : all_reduce on the capture stream            fork : on a side stream forked from / joined to the capture stream
  async  : fork + async_op=True / work.wait()"""
import faulthandler
import os
import sys

faulthandler.enable()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyedm_amd  # noqa: E402,F401
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "same"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29542")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
y = torch.zeros(1 << 20, device=dev)
cap_stream, comm = torch.cuda.Stream(), torch.cuda.Stream()


def body():
    y.copy_(x * 2)
    if mode == "same":
        dist.all_reduce(y)
    else:
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            if mode == "async":
                w = dist.all_reduce(y, async_op=True)
                w.wait()
            else:
                dist.all_reduce(y)
        torch.cuda.current_stream().wait_stream(comm)
    y.add_(1)


cap_stream.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap_stream):
    for _ in range(3):
        body()
torch.cuda.synchronize()
print(mode, "eager ok", float(y[0]), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap_stream):
    body()
print(mode, "captured", flush=True)
x.fill_(5)
g.replay()
torch.cuda.synchronize()
print(mode, "replayed", float(y[0]), "(expect 11)", flush=True)
dist.destroy_process_group()
print(mode, "OK", flush=True)
