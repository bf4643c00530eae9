"""The attention block's kernel sequences on the CIFAR-10 shapes, unfused (1x1 conv + attention.hip + 1x1 conv) against
fused (csrc/attention_fused.hip + out conv), with ROTATING operand sets so that the 256 MB Infinity Cache does not serve
the re-reads:  python tools/microbench_attnblock.py [hp]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyedm_amd import ops  # noqa: E402

bf16 = torch.bfloat16


def timed(fn, sets, iters=40):
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def run(B, H, W, heads=4, C=256, nsets=12):
    sets = []
    for _ in range(nsets):
        x = torch.randn(B, H, W, C, device="cuda").to(bf16)
        gout = torch.randn(B, H, W, C, device="cuda").to(bf16)
        wf = (torch.randn(1, 3 * C, C, device="cuda") / 16).to(bf16)
        wd_qkv = (torch.randn(1, C, 3 * C, device="cuda") / 16).to(bf16)
        wo = (torch.randn(1, C, C, device="cuda") / 16).to(bf16)
        wdo = (torch.randn(1, C, C, device="cuda") / 16).to(bf16)
        qkv = ops.conv_igemm(x, wf, 1)
        y = ops.attention_fwd(qkv, heads)
        y2, stat = ops.attention_qkv_fwd(x, wf, heads)
        gy = ops.conv_igemm(gout, wdo, 1, alpha=0.7)
        sets.append(dict(x=x, gout=gout, wf=wf, wd_qkv=wd_qkv, wo=wo, wdo=wdo, qkv=qkv, y=y, y2=y2, stat=stat, gy=gy))
    t = {}
    t["qkv conv"] = timed(lambda s: ops.conv_igemm(s["x"], s["wf"], 1), sets)
    t["attn fwd"] = timed(lambda s: ops.attention_fwd(s["qkv"], heads), sets)
    t["FUSED fwd"] = timed(lambda s: ops.attention_qkv_fwd(s["x"], s["wf"], heads), sets)
    t["out conv+mp_add"] = timed(lambda s: ops.conv_igemm(s["y"], s["wo"], 1, residual=s["x"], alpha=0.7, beta=0.7), sets)
    t["dgrad out"] = timed(lambda s: ops.conv_igemm(s["gout"], s["wdo"], 1, alpha=0.7), sets)
    t["attn bwd"] = timed(lambda s: ops.attention_bwd(s["qkv"], s["y"], s["gy"], heads), sets)
    t["FUSED bwd"] = timed(lambda s: ops.attention_qkv_bwd(s["x"], s["y2"], s["gout"], s["stat"], s["wf"], s["wdo"], heads, 0.7), sets)
    gq = ops.attention_bwd(sets[0]["qkv"], sets[0]["y"], sets[0]["gy"], heads)
    t["dgrad qkv"] = timed(lambda s: ops.conv_igemm(gq, s["wd_qkv"], 1, residual=s["gout"], alpha=1.0, beta=0.7), sets)
    print(f"B={B} {H}x{W} (ATTN_HP={ops.ATTN_HP}):  " + "  ".join(f"{k} {v:6.1f}" for k, v in t.items()), flush=True)
    print(f"    forward  unfused {t['qkv conv'] + t['attn fwd'] + t['out conv+mp_add']:6.1f} us   fused "
          f"{t['FUSED fwd'] + t['out conv+mp_add']:6.1f} us", flush=True)
    print(f"    backward unfused {t['dgrad out'] + t['attn bwd'] + t['dgrad qkv']:6.1f} us   fused "
          f"{t['FUSED bwd'] + t['dgrad qkv']:6.1f} us", flush=True)


if __name__ == "__main__":
    for hp in ([int(a) for a in sys.argv[1:]] or [0]):
        ops.ATTN_HP = hp
        run(128, 16, 16)
        run(128, 8, 8)
        run(512, 16, 16, nsets=4)
