"""Golden vectors for single Encoder/Decoder blocks at widths the HIP kernels support (64 channels, head dim 32,
8x8 maps), from the *reference's own* modules (build container only; TEST INFRASTRUCTURE ONLY).  The round-1
`blocks.npz` used 16/32-channel blocks with head dim 16, which the HIP convs (C % 32) and attention (d in
{32,64,128,144,192}) do not cover, so `enc_attn / enc_widen / dec_skip / dec_skip_attn` never ran on the GPU.

Parameters are NOT stored: both sides regenerate them from the seed with the CPU generator, in sorted-name order
(`block_params`); a digest of every tensor is stored so a silent RNG change is detected.

Usage:  python oracle/make_golden_blocks64.py   ->  tests/golden/blocks64.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import _load, grad_digest  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
EMB = 64

#        tag              kind   cin cout skip down/up attn  hw  seed
SPECS = [("enc_plain",     "enc", 64, 64,  0,  False, False, 8, 201),
         ("enc_down",      "enc", 64, 64,  0,  True,  False, 8, 202),
         ("enc_attn",      "enc", 64, 64,  0,  False, True,  8, 203),
         ("enc_widen",     "enc", 32, 64,  0,  False, False, 8, 204),
         ("dec_plain",     "dec", 64, 64,  0,  False, False, 8, 205),
         ("dec_up",        "dec", 64, 64,  0,  True,  False, 4, 206),
         ("dec_skip_attn", "dec", 64, 64,  64, False, True,  8, 207),
         ("dec_skip",      "dec", 64, 32,  64, False, False, 8, 208),
         ("dec_skip_up",   "dec", 64, 64,  32, True,  False, 4, 209)]


def block_params(module, seed):
    """seeded parameters in sorted-name order (scalar gains: 1.1): the same call on the reference module and on the
    tinyedm_amd module yields the same tensors because the two share parameter names and shapes"""
    gg = torch.Generator().manual_seed(seed)
    out = {}
    for name, p in sorted(module.named_parameters(), key=lambda kv: kv[0]):
        out[name] = torch.randn(p.shape, generator=gg) if p.ndim else torch.tensor(1.1)
    return out


def block_inputs(cin, skip, hw, seed):
    gg = torch.Generator().manual_seed(seed + 5000)
    x = torch.randn(2, cin, hw, hw, generator=gg)
    s = torch.randn(2, skip, hw, hw, generator=gg) if skip else None
    return x, s


def main():
    net = _load("networks")
    blk = {}
    embv = torch.randn(2, EMB, generator=torch.Generator().manual_seed(21))
    blk["emb"] = embv.numpy()
    for tag, kind, cin, cout, skip, resample, attn, hw, seed in SPECS:
        if kind == "enc":
            m = net.EncoderBlock(cin, cout, EMB, resample, attn, num_heads=2).eval()
        else:
            m = net.DecoderBlock(cin, cout, EMB, resample, attn, num_heads=2, skip_channels=skip).eval()
        P = block_params(m, seed)
        m.load_state_dict(P, strict=True)
        x, s = block_inputs(cin, skip, hw, seed)
        with torch.no_grad():
            y32 = m(x, embv) if kind == "enc" else m(x, embv, s)
            # the reference's own bf16 policy (cpu autocast stands in for cuda); block inputs are bf16 inside the
            # network (they come out of a conv), which is also what keeps the reference's lerp dtypes consistent
            xb = x.to(torch.bfloat16)
            sb = None if s is None else s.to(torch.bfloat16)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                ybf = m(xb, embv) if kind == "enc" else m(xb, embv, sb)
        blk[tag + "::y"] = y32.numpy()
        blk[tag + "::y_autocast_bf16"] = ybf.float().numpy()
        blk[tag + "::digest"] = np.stack([grad_digest(P[k]) for k in sorted(P)])
        blk[tag + "::keys"] = np.array(sorted(P))
    np.savez_compressed(os.path.join(OUT, "blocks64.npz"), **blk)
    print("wrote", os.path.join(OUT, "blocks64.npz"), os.path.getsize(os.path.join(OUT, "blocks64.npz")), "bytes")


if __name__ == "__main__":
    main()
