"""The fp32 Linear GEMMs of a step / a sampler evaluation (csrc/linear.hip k_sgemm_mfma): us per call.
    python tools/microbench_linear.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

dev = "cuda"


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for B in (128, 512):
    emb = torch.randn(B, 256, device=dev)
    wcat = torch.randn(5376, 256, device=dev)
    glin = torch.randn(B, 5376, device=dev)
    four = torch.randn(B, 64, device=dev)
    ws = torch.randn(256, 64, device=dev)
    ges = torch.randn(B, 256, device=dev)
    print(f"B={B}: embed-all fwd ({B}x256 -> 5376) {timed(lambda: ops.linear_fwd(emb, wcat)):6.1f} us   "
          f"wgrad (5376x256 over {B}) {timed(lambda: ops.linear_wgrad(glin, emb)):6.1f} us   "
          f"dgrad ({B}x5376 -> 256, split-K) {timed(lambda: ops.linear_dgrad(glin, wcat)):6.1f} us   "
          f"sigma embed fwd {timed(lambda: ops.linear_fwd(four, ws)):6.1f} us   its wgrad {timed(lambda: ops.linear_wgrad(ges, four)):6.1f} us",
          flush=True)
