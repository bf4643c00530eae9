"""Per-block error growth of the bf16 evaluation path on the CIFAR-10 net: after every encoder / decoder
block, the relative L2 distance of the bf16 HIP activations from the reference-precision (exact-fp32, csrc/eval_f32.hip)
activations of the same network on the same input -- the fp32 path itself sits 3e-6 from the fp32 CPU oracle
(tests/test_evalf32_gpu.py).  A per-kernel rounding of 1.65e-3 (one bf16 rounding of each output) accumulating like a
random walk over the ~3 roundings per block predicts err(depth d) ~ 1.65e-3 * sqrt(3 d): the table shows measured vs that.
    python tools/error_growth.py [--batch 16] > profiles/r03_error_growth.json"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyedm_amd  # noqa: E402,F401
import torch  # noqa: E402

from oracle import edm_oracle as O  # noqa: E402  (parameter initialiser only: tools may use the oracle)
import tinyedm_amd as T  # noqa: E402
from tinyedm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--conditional", action="store_true")
a = ap.parse_args()
dev = "cuda"
ecfg, dcfg = O.cifar10_cfg(10 if a.conditional else None)
P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(1), gains_nonzero=True)
emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types), tuple(dcfg.decoder_block_types),
                 tuple(dcfg.encoder_out_channels), tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections),
                 dcfg.dropout_rate, dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim,
                 dcfg.num_heads)
emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
emb, den = emb.to(dev).eval(), den.to(dev).eval()
g = torch.Generator().manual_seed(12)
B = a.batch
clean = 0.5 * torch.randn(B, 3, 32, 32, generator=g)
sigma = (torch.randn(B, generator=g) * 1.2 - 1.2).exp()
noisy = (clean + sigma.view(-1, 1, 1, 1) * torch.randn(B, 3, 32, 32, generator=g)).to(dev)
labels = torch.randint(0, 10, (B,), generator=g).to(dev) if a.conditional else None
sigma = sigma.to(dev)

blocks = [(f"enc{i}:{t}", b) for i, (b, t) in enumerate(zip(den.encoder_blocks, den.encoder_block_types))] + \
         [(f"dec{i}:{t}", b) for i, (b, t) in enumerate(zip(den.decoder_blocks, den.decoder_block_types))]
rec = {"bf16": {}, "f32": {}}


def wrap(name, blk):
    f_b, f_f = blk.forward, blk.forward_f32

    def fwd(*args, **kw):
        out = f_b(*args, **kw)
        rec["bf16"][name] = (out[0] if isinstance(out, tuple) else out).float()      # NHWC bf16 -> fp32
        return out

    def fwd32(*args, **kw):
        out = f_f(*args, **kw)
        rec["f32"][name] = out
        return out
    blk.forward, blk.forward_f32 = fwd, fwd32


for name, blk in blocks:
    wrap(name, blk)
with torch.no_grad():
    _, e = emb(sigma, labels)
    den.set_eval_dtype("bf16")
    D_b = den(noisy, sigma, e)
    den.set_eval_dtype("f32")
    D_f = den(noisy, sigma, e)


def rel(x, y):
    return ((x.double() - y.double()).norm() / y.double().norm()).item()


rows = []
for d, (name, _) in enumerate(blocks, 1):
    rows.append({"block": name, "depth": d, "rel_err_bf16_vs_f32": rel(rec["bf16"][name], rec["f32"][name]),
                 "random_walk_model": 1.65e-3 * math.sqrt(3 * d)})
c_skip = (0.25 / (sigma ** 2 + 0.25)).view(-1, 1, 1, 1)
out = {"config": "cifar10_cond" if a.conditional else "cifar10", "batch": B,
       "what": "relative L2 error of the bf16 HIP activations against the exact-fp32 HIP activations after every block "
               "(eval mode, same weights and input); random_walk_model = 1.65e-3 * sqrt(3 * depth)",
       "blocks": rows, "output_D_minus_cskip_x": rel(D_b - c_skip * noisy, D_f - c_skip * noisy)}
print(json.dumps(out, indent=1))
for r in rows:
    print(f"# {r['block']:10s} depth {r['depth']:2d}  measured {r['rel_err_bf16_vs_f32']:.3e}  model {r['random_walk_model']:.3e}",
          file=sys.stderr)
print(f"# network output (D - c_skip x): {out['output_D_minus_cskip_x']:.3e}", file=sys.stderr)
