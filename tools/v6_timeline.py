"""Per-workgroup timeline of k_conv3x3_v6 (diagnostic build: hipcc -DEDM_V6_TIMELINE, library given by EDM_LIB_PATH):
when each workgroup starts, how long its prologue (first slab + weight tiles landed), main loop, epilogue and store drain
take, and how long a CU sits between two workgroups.  s_memrealtime ticks (10 ns).
    EDM_LIB_PATH=$PWD/ab/lib_timeline.so python tools/v6_timeline.py [B] [Cin] [Cout] [HW]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops, _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 256
HW = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = "cuda"
x = torch.randn(B, HW, HW, Cin, device=dev).to(torch.bfloat16)
wp = (torch.randn(9, Cout, Cin, device=dev) / (Cin * 9) ** 0.5).to(torch.bfloat16)
for _ in range(5):
    ops.conv_igemm(x, wp, 9)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
nwg = 8192
buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
assert h.edm_v6_set_timeline(ctypes.c_void_p(buf.data_ptr())) == 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    buf.zero_()
    torch.cuda.synchronize()
    s.record()
    ops.conv_igemm(x, wp, 9)
    e.record()
    torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 8)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
us = lambda v: (v - t0) / 100.0
print(f"B={B} {HW}x{HW} {Cin}->{Cout}: {len(t)} workgroups, event time {s.elapsed_time(e) * 1e3:.1f} us, "
      f"first start -> last store retired {us(t[:, 4].max()):.1f} us")
ph = {"prologue": t[:, 1] - t[:, 0], "main loop": t[:, 2] - t[:, 1], "epilogue issue": t[:, 3] - t[:, 2], "store drain": t[:, 4] - t[:, 3]}
cu = t[:, 5]
order = np.argsort(t[:, 0])
first = np.zeros(len(t), bool)
seen = {}
gaps = []
for i in order:
    k = int(cu[i]) & ~0xFF | (int(cu[i]) >> 32) << 40     # (xcc, se/sh/cu): drop wave / simd / pipe bits
    if k not in seen:
        first[i] = True
    else:
        gaps.append((t[i, 0] - t[seen[k], 4]) / 100.0)
    seen[k] = i
print(f"CUs seen: {len(seen)}; start of the first-round workgroups: median {np.median(us(t[first, 0])):.1f} us, max {us(t[first, 0]).max():.1f} us")
for name, v in ph.items():
    a, b = v[first] / 100.0, v[~first] / 100.0
    print(f"{name:15s} round 1: median {np.median(a):6.1f} (p10 {np.percentile(a, 10):6.1f}, p90 {np.percentile(a, 90):6.1f}) us"
          + (f"   later rounds: median {np.median(b):6.1f} (p10 {np.percentile(b, 10):6.1f}, p90 {np.percentile(b, 90):6.1f}) us" if len(b) else ""))
if gaps:
    g = np.array(gaps)
    print(f"CU idle between a workgroup's last store and the next workgroup's start: median {np.median(g):.1f} us (p10 {np.percentile(g, 10):.1f}, p90 {np.percentile(g, 90):.1f})")
clk = (t[:, 7] - t[:, 6]) / np.maximum(t[:, 2] - t[:, 1], 1) * 100.0       # MHz: shader-clock ticks per 10-ns tick
steps = Cin // 32 * 9
mf = steps * 32 * 2 * 16        # MFMA cycles of a SIMD's two waves over the main loop (32 per step and wave, 16 cycles each)
print(f"shader clock inside the main loop (s_memtime / s_memrealtime): median {np.median(clk):.0f} MHz (p10 {np.percentile(clk, 10):.0f}, "
      f"p90 {np.percentile(clk, 90):.0f}); the loop's {mf} MFMA cycles per SIMD would take {mf / np.median(clk):.1f} us at that clock: "
      f"matrix-pipe occupancy {mf / np.median(clk) / np.median((t[:, 2] - t[:, 1]) / 100.0) * 100:.0f} %")
end = us(t[:, 4])
print(f"workgroup end times: p10 {np.percentile(end, 10):.1f}  median {np.median(end):.1f}  p90 {np.percentile(end, 90):.1f}  max {end.max():.1f} us")
last = np.array([us(t[i, 4]) for i in seen.values()])
print(f"per-CU finish: min {last.min():.1f}  median {np.median(last):.1f}  max {last.max():.1f} us")
