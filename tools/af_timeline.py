"""Per-WAVE timeline of the fused attention kernels (diagnostic build -DEDM_AF_TIMELINE, library given by EDM_LIB_PATH);
s_memrealtime ticks (10 ns).
    python tools/build_diag_lib.py ab/lib_af_timeline.so -DEDM_AF_TIMELINE
    EDM_LIB_PATH=$PWD/ab/lib_af_timeline.so python tools/af_timeline.py [B] [HW] [hp]    (hp: heads per workgroup of the FORWARD kernel only)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops, _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ops.ATTN_HP = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev, C, heads = "cuda", 256, 4
bf16 = torch.bfloat16
x = torch.randn(B, HW, HW, C, device=dev).to(bf16)
gout = torch.randn(B, HW, HW, C, device=dev).to(bf16)
wf = (torch.randn(1, 3 * C, C, device=dev) / 16).to(bf16)
wdo = (torch.randn(1, C, C, device=dev) / 16).to(bf16)
y, stat = ops.attention_qkv_fwd(x, wf, heads)
h = ctypes.CDLL(_lib.LIB_PATH)
nw = 2048 * 8
buf = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
assert h.edm_af_set_timeline(ctypes.c_void_p(buf.data_ptr())) == 0


def show(name, fn, phases, last):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    buf.zero_()
    torch.cuda.synchronize()
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nw, 8)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    print(f"{name} B={B} {HW}x{HW}: {len(t)} waves, event time {s.elapsed_time(e) * 1e3:.1f} us, first start -> last wave done "
          f"{(t[:, last].max() - t0) / 100.0:.1f} us; starts spread over {(t[:, 0].max() - t0) / 100.0:.1f} us")
    for label, a, b in phases:
        v = (t[:, b] - t[:, a]) / 100.0
        print(f"  {label:44s} median {np.median(v):6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")


show("k_attn_qkv_fwd", lambda: ops.attention_qkv_fwd(x, wf, heads),
     [("x fragments + qkv projection (head 0)", 0, 1), ("norm + K/V image writes + barrier", 1, 2),
      ("streamed attention (head 0)", 2, 3), ("stores + remaining heads", 3, 4), ("whole wave", 0, 4)], 4)
show("k_attn_qkv_bwd", lambda: ops.attention_qkv_bwd(x, y, gout, stat, wf, wdo, heads, 0.7),
     [("loads + dO projection (phase B)", 0, 1), ("qkv projection (phase A)", 1, 2), ("norm + images + barriers", 2, 3),
      ("pass 1 (dQ)", 3, 4), ("pass 2 (dK, dV) up to the last store", 4, 5), ("final stores", 5, 6), ("whole wave", 0, 6)], 6)
