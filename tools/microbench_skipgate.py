"""ScaleLong gate kernels (k_skip_gate_fwd / _bwd) at the three resolutions of the CIFAR-10 config, B = 128, hipGraph-replayed
with rotating operands (no Infinity-Cache hits from the previous iteration).  python tools/microbench_skipgate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

B, C, R = 128, 256, 64
dev = "cuda"
for HW in (32, 16, 8):
    NROT = 6
    skips = [torch.randn(B, HW, HW, C, device=dev).to(torch.bfloat16) for _ in range(NROT)]
    gcats = [torch.randn(B, HW, HW, 2 * C, device=dev).to(torch.bfloat16) for _ in range(NROT)]
    w1 = torch.randn(R, C + 1, device=dev) / 16
    w2 = torch.randn(C, R, device=dev) / 8
    mean, gate, z1 = ops.skip_gate_fwd(skips[0], w1, w2)
    res = {}
    for name, fn in (("fwd", lambda i: ops.skip_gate_fwd(skips[i], w1, w2)),
                     ("bwd", lambda i: ops.skip_gate_bwd(gcats[i], C, skips[i], mean, w1, w2, gate, z1))):
        for i in range(NROT):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(NROT):
                fn(i)
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        res[name] = s.elapsed_time(e) / (10 * NROT) * 1e3
    print(f"{HW:2d}x{HW:<2d}: fwd {res['fwd']:6.1f} us   bwd {res['bwd']:6.1f} us ", flush=True)
