"""Generate golden fixtures from the *reference itself* (runs only in the build
container, where /root/reference exists).  TEST INFRASTRUCTURE ONLY.

The reference's ``networks.py`` / ``solvers.py`` / ``utils.py`` depend only on
torch + numpy, so they are loaded standalone with importlib (bypassing
``tinyedm/__init__.py``, which pulls lightning).  ``edm.py`` / ``ema.py`` /
``metric.py`` are not importable here (lightning / torchmetrics absent); for
those the fixtures hold the closed-form known-answer values instead.

Outputs (small, data only -- inputs and expected outputs):
    tests/golden/tiny_net.npz     tiny Denoiser+Embedding: eval fwd, train fwd, grads
    tests/golden/blocks.npz       single Encoder/Decoder blocks of every flavour
    tests/golden/ops.npz          L0 ops fwd
    tests/golden/solver.npz       t_steps tables (bitwise) + a Heun trajectory
    tests/golden/tables.npz       default architecture tables

Weights are *not* stored: they are regenerated from a seed through
``oracle.edm_oracle.init_params`` (torch CPU RNG is reproducible for a fixed
torch build, and the GPU box runs the same image).  A per-tensor checksum of
the generated parameters is stored so a silent RNG change is detected.

Usage:  python oracle/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edm_oracle as O  # noqa: E402

REF = "/root/reference/src/tinyedm"
OUT = os.path.join(ROOT, "tests", "golden")


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def tiny_cfgs(num_classes=10):
    e = O.EmbeddingCfg(fourier_dim=32, embedding_dim=64, num_classes=num_classes)
    d = O.DenoiserCfg(
        in_channels=3, out_channels=3,
        encoder_block_types=["Enc", "EncD", "EncA"],
        decoder_block_types=["DecA", "DecA", "DecU", "Dec", "Dec"],
        encoder_out_channels=[64, 128, 128],
        decoder_out_channels=[128, 128, 128, 64, 64],
        skip_connections=[True, True, False, True, True],
        dropout_rate=0.0, sigma_data=0.5, embedding_dim=64, num_heads=2)
    return e, d


def build_ref(net, ecfg, dcfg, P):
    emb = net.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = net.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                       tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                       tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), dcfg.dropout_rate,
                       dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor,
                       dcfg.embedding_dim, dcfg.num_heads)
    sd_e = {k[len("embedding."):]: v.clone() for k, v in P.items() if k.startswith("embedding.")}
    sd_d = {k[len("denoiser."):]: v.clone() for k, v in P.items() if k.startswith("denoiser.")}
    emb.load_state_dict(sd_e, strict=True)
    den.load_state_dict(sd_d, strict=True)
    return emb, den


def grad_digest(g):
    g = g.detach().double().flatten()
    return np.array([g.sum().item(), g.norm().item()] + g[:14].tolist() + [0.0] * max(0, 14 - g.numel()))[:16]


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    net, sol = _load("networks"), _load("solvers")

    # ---- architecture tables (tests/test_unet_builder.py:14-30) ----
    np.savez(os.path.join(OUT, "tables.npz"),
             enc_types=np.array(net.get_encoder_blocks_types()),
             dec_types=np.array(net.get_decoder_blocks_types()),
             enc_ch=np.array(net.get_encoder_out_channels()),
             dec_ch=np.array(net.get_decoder_out_channels()),
             skips=np.array(net.get_skip_connections()),
             skip_ch=np.array(net.get_skip_channels(net.get_encoder_out_channels(), net.get_decoder_out_channels(),
                                                    net.get_skip_connections())))

    # ---- L0 ops ----
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 64, 4, 4, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g)
    a, b = torch.randn(2, 8, generator=g), torch.randn(2, 8, generator=g)
    conv = net.Conv2d(64, 32, 3).eval()
    conv.weight.data.copy_(w)
    lin = net.Linear(64, 32).eval()
    lin.weight.data.copy_(w[:, :, 0, 0])
    with torch.no_grad():
        np.savez(os.path.join(OUT, "ops.npz"), x=x.numpy(), w=w.numpy(), a=a.numpy(), b=b.numpy(),
                 pixel_norm=net.pixel_norm(x).numpy(), normalize=net.normalize(w).numpy(),
                 conv=conv(x).numpy(), linear=lin(x[:, :, 0, 0]).numpy(),
                 mp_silu=net.mp_silu(x).numpy(), mp_add03=net.mp_add(a, b, 0.3).numpy(),
                 mp_add05=net.mp_add(a, b).numpy(),
                 up=net.UpSample()(x).numpy(), down=net.DownSample()(x).numpy())

    # ---- tiny network: eval forward, training forward (+side effect), grads ----
    ecfg, dcfg = tiny_cfgs()
    gen = torch.Generator().manual_seed(7)
    P = O.init_params(ecfg, dcfg, gen)
    gi = torch.Generator().manual_seed(8)
    B = 4
    clean = 0.5 * torch.randn(B, 3, 8, 8, generator=gi)
    eps = torch.randn(B, generator=gi)
    noise = torch.randn(B, 3, 8, 8, generator=gi)
    labels = torch.randint(0, 10, (B,), generator=gi)
    noisy, sigma = O.diffuse(clean, eps, noise, -1.2, 1.2)

    emb_m, den_m = build_ref(net, ecfg, dcfg, P)
    out = {"clean": clean.numpy(), "eps": eps.numpy(), "noise": noise.numpy(), "labels": labels.numpy(),
           "noisy": noisy.numpy(), "sigma": sigma.numpy()}
    out["param_digest"] = np.stack([grad_digest(P[k]) for k in sorted(P)])
    out["param_keys"] = np.array(sorted(P))

    emb_m.eval(); den_m.eval()
    with torch.no_grad():
        four, e = emb_m(sigma, labels)
        out["eval_fourier"], out["eval_emb"] = four.numpy(), e.numpy()
        out["eval_D"] = den_m(noisy, sigma, e).numpy()
        _, e_u = emb_m(sigma, None)
        out["eval_D_uncond"] = den_m(noisy, sigma, e_u).numpy()
        # 0-dim sigma + (B,1) labels broadcast, as the solver calls it (solvers.py:48)
        s0 = torch.tensor(1.7)
        _, e0 = emb_m(s0, labels.view(-1, 1))
        out["eval_D_scalar_sigma"] = den_m(noisy, s0, e0).numpy()
        # reference under its own bf16 autocast policy (cpu autocast stands in for cuda)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            _, eb = emb_m(sigma, labels)
            out["eval_D_autocast_bf16"] = den_m(noisy, sigma, eb).float().numpy()

    emb_m.train(); den_m.train()
    _, e = emb_m(sigma, labels)
    D = den_m(noisy, sigma, e)
    weight = (sigma ** 2 + 0.5 ** 2) / (sigma * 0.5) ** 2
    loss = torch.mean(weight.view(-1, 1, 1, 1) * (D - clean) ** 2)  # closed form, tests/test_weighted_mean_squared_error.py:18-21
    loss.backward()
    out["train_D"] = D.detach().numpy()
    out["train_loss"] = np.array(loss.item())
    named = {("embedding." + k): v for k, v in emb_m.named_parameters()}
    named.update({("denoiser." + k): v for k, v in den_m.named_parameters()})
    keys = sorted(named)
    out["grad_keys"] = np.array(keys)
    out["grad_digest"] = np.stack([grad_digest(named[k].grad) for k in keys])
    out["post_param_digest"] = np.stack([grad_digest(named[k]) for k in keys])
    for k in ("denoiser.gain_out", "denoiser.conv_out.weight", "denoiser.encoder_blocks.0.gain",
              "denoiser.decoder_blocks.0.cat_factor.layer1.weight", "embedding.class_embed.linear.weight"):
        out["grad::" + k] = named[k].grad.numpy()
    np.savez(os.path.join(OUT, "tiny_net.npz"), **out)

    # ---- single blocks of each flavour ----
    blk = {}
    gb = torch.Generator().manual_seed(21)
    embv = torch.randn(2, 64, generator=gb)
    blk["emb"] = embv.numpy()

    def run_enc(tag, seed, cin, cout, down, attn, hw):
        m = net.EncoderBlock(cin, cout, 64, down, attn, num_heads=2).eval()
        gg = torch.Generator().manual_seed(seed)
        for n_, p_ in m.named_parameters():
            p_.data.copy_(torch.randn(p_.shape, generator=gg) if p_.ndim else torch.tensor(1.3))
        xin = torch.randn(2, cin, hw, hw, generator=gg)
        blk[tag + "::x"] = xin.numpy()
        for n_, p_ in m.named_parameters():
            blk[tag + "::p::" + n_] = p_.detach().numpy()
        with torch.no_grad():
            blk[tag + "::y"] = m(xin, embv).numpy()

    def run_dec(tag, seed, cin, cout, sc, up, attn, hw):
        m = net.DecoderBlock(cin, cout, 64, up, attn, num_heads=2, skip_channels=sc).eval()
        gg = torch.Generator().manual_seed(seed)
        for n_, p_ in m.named_parameters():
            p_.data.copy_(torch.randn(p_.shape, generator=gg) if p_.ndim else torch.tensor(0.8))
        xin = torch.randn(2, cin, hw, hw, generator=gg)
        blk[tag + "::x"] = xin.numpy()
        skip = torch.randn(2, sc, hw, hw, generator=gg) if sc else None
        if sc:
            blk[tag + "::skip"] = skip.numpy()
        for n_, p_ in m.named_parameters():
            blk[tag + "::p::" + n_] = p_.detach().numpy()
        with torch.no_grad():
            blk[tag + "::y"] = m(xin, embv, skip).numpy()

    run_enc("enc_plain", 101, 32, 32, False, False, 4)
    run_enc("enc_down", 102, 32, 32, True, False, 4)
    run_enc("enc_attn", 103, 32, 32, False, True, 4)
    run_enc("enc_widen", 104, 16, 32, False, False, 4)
    run_dec("dec_plain", 105, 32, 32, 0, False, False, 4)
    run_dec("dec_up", 106, 32, 32, 0, True, False, 4)
    run_dec("dec_skip_attn", 107, 32, 32, 32, False, True, 4)
    run_dec("dec_skip", 108, 32, 16, 32, False, False, 4)
    np.savez(os.path.join(OUT, "blocks.npz"), **blk)

    # ---- solver: tables bitwise + trajectory with the tiny net as the model ----
    s18, s32 = sol.DeterministicSolver(18), sol.DeterministicSolver(32)
    s5 = sol.DeterministicSolver(5, sigma_min=0.01, sigma_max=20.0, rho=5.0)
    emb_m.eval(); den_m.eval()

    def model(xx, t, lab):
        _, e_ = emb_m(t, lab)
        return den_m(xx, t, e_)

    gs = torch.Generator().manual_seed(99)
    x0 = torch.randn(2, 3, 8, 8, generator=gs)
    lab = torch.tensor([[3], [7]])
    with torch.no_grad():
        traj5 = s5.solve(model, x0, lab)
        traj18 = s18.solve(model, x0, None)
    np.savez(os.path.join(OUT, "solver.npz"), t18=s18.t_steps.numpy(), t32=s32.t_steps.numpy(),
             t5=s5.t_steps.numpy(), x0=x0.numpy(), labels=lab.numpy(), x_heun5=traj5.numpy(),
             x_heun18_uncond=traj18.numpy())
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
