"""Per-block parity of the bf16 HIP path against the bf16-rounding CPU ORACLE (not against fp32: tools/error_growth.py does
that): after every encoder / decoder block of the CIFAR-10 net, the relative L2 distance of the HIP activations from
oracle.edm_oracle.denoiser_forward(bf16=True)'s on the same input -- two independent bf16 evaluations of the same network
(each ~8e-3 from fp32 by the end) -- and from the FP32 oracle for scale.  The end-to-end limits of the GPU tests (2.5e-2) are
set from the last rows of this table with the margin printed at the bottom.
    python tools/block_parity.py [--batch 4] > profiles/r04_block_parity.json"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyedm_amd as T  # noqa: E402
import torch  # noqa: E402

from oracle import edm_oracle as O  # noqa: E402  (tools may use the oracle)
from tinyedm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = "cuda"
ecfg, dcfg = O.cifar10_cfg(None)
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
noisy = clean + sigma.view(-1, 1, 1, 1) * torch.randn(B, 3, 32, 32, generator=g)

names = [f"denoiser.encoder_blocks.{i}." for i in range(len(den.encoder_blocks))] + \
        [f"denoiser.decoder_blocks.{i}." for i in range(len(den.decoder_blocks))]
blocks = list(den.encoder_blocks) + list(den.decoder_blocks)
hip = {}


def wrap(name, blk):
    f = blk.forward

    def fwd(*args, **kw):
        out = f(*args, **kw)
        o = out[0] if isinstance(out, tuple) else out
        hip[name] = ops.nhwc_bf16_to_nchw(o.contiguous() if o.is_contiguous() else o.clone().contiguous()).cpu()
        return out
    blk.forward = fwd


for n_, b_ in zip(names, blocks):
    wrap(n_, b_)
with torch.no_grad():
    _, e = emb(sigma.to(dev), None)
    D = den(noisy.to(dev), sigma.to(dev), e).cpu()
    _, e_or = O.embedding_forward(P, ecfg, sigma, None)
    rb, rf = {}, {}
    D_b = O.denoiser_forward(P, dcfg, noisy, sigma, e_or, False, True, record=rb)
    D_f = O.denoiser_forward(P, dcfg, noisy, sigma, e_or, False, False, record=rf)


def rel(x, y):
    return ((x.double() - y.double()).norm() / y.double().norm()).item()


rows = []
for d, n_ in enumerate(names, 1):
    rows.append({"block": n_[len("denoiser."):-1], "depth": d, "hip_bf16_vs_bf16_oracle": rel(hip[n_], rb[n_]),
                 "hip_bf16_vs_fp32_oracle": rel(hip[n_], rf[n_]), "bf16_oracle_vs_fp32_oracle": rel(rb[n_], rf[n_])})
c_skip, _, _ = O.precond_scalars(sigma, dcfg.sigma_data)
base = c_skip * noisy
end = rel(D - base, D_b - base)
out = {"config": "cifar10", "batch": B,
       "what": "relative L2 error after every block (eval mode, same weights and input): the bf16 HIP activations against the "
               "bf16-rounding CPU oracle and against the fp32 oracle, and the bf16 oracle against the fp32 oracle",
       "blocks": rows, "output_D_minus_cskip_x_hip_vs_bf16_oracle": end, "test_limit": 2.5e-2, "margin": 2.5e-2 / end}
print(json.dumps(out, indent=1))
for r in rows:
    print(f"# {r['block']:20s} HIP vs bf16 oracle {r['hip_bf16_vs_bf16_oracle']:.3e}   HIP vs fp32 {r['hip_bf16_vs_fp32_oracle']:.3e}   "
          f"bf16 oracle vs fp32 {r['bf16_oracle_vs_fp32_oracle']:.3e}", file=sys.stderr)
print(f"# network output (D - c_skip x), HIP vs bf16 oracle: {end:.3e}; test limit 2.5e-2 -> margin {2.5e-2 / end:.1f}", file=sys.stderr)
