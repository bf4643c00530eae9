"""Per-WAVE timeline of k_conv3x3_s (diagnostic build -DEDM_S_TIMELINE, library given by EDM_LIB_PATH): start, first
fragments landed, end of the first round, end of the main loop, reduction done, stores retired.  s_memrealtime ticks (10 ns).
    python tools/build_diag_lib.py ab/lib_s_timeline.so -DEDM_S_TIMELINE
    EDM_LIB_PATH=$PWD/ab/lib_s_timeline.so python tools/s_timeline.py [B] [Cin] [Cout] [HW]"""
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
HW = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = "cuda"
x = torch.randn(B, HW, HW, Cin, device=dev).to(torch.bfloat16)
wp = (torch.randn(9, Cout, Cin, device=dev) / (Cin * 9) ** 0.5).to(torch.bfloat16)
assert ops._igemm_entry(B * HW * HW, HW, Cout, 9, Cin) == "edm_conv_igemm_s"
for _ in range(5):
    ops.conv_igemm(x, wp, 9)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
nw = 4096 * 4
buf = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
assert h.edm_s_set_timeline(ctypes.c_void_p(buf.data_ptr())) == 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    buf.zero_()
    torch.cuda.synchronize()
    s.record()
    ops.conv_igemm(x, wp, 9)
    e.record()
    torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nw, 8)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
print(f"B={B} {HW}x{HW} {Cin}->{Cout}: {len(t)} waves, event time {s.elapsed_time(e) * 1e3:.1f} us, first start -> last store "
      f"retired {(t[:, 4].max() - t0) / 100.0:.1f} us; starts spread over {(t[:, 0].max() - t0) / 100.0:.1f} us")
ph = {"set-up (loads issued, addresses formed)": t[:, 6] - t[:, 0], "wait for the slab": t[:, 1] - t[:, 6], "round 0 (9 steps)": t[:, 5] - t[:, 1],
      "rest of the main loop": t[:, 2] - t[:, 5], "reduction through LDS": t[:, 3] - t[:, 2], "epilogue + store drain": t[:, 4] - t[:, 3],
      "whole wave": t[:, 4] - t[:, 0]}
for name, v in ph.items():
    a = v / 100.0
    print(f"{name:36s} median {np.median(a):6.2f}  p10 {np.percentile(a, 10):6.2f}  p90 {np.percentile(a, 90):6.2f}  max {a.max():6.2f} us")
