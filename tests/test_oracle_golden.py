"""Pin the CPU oracle (oracle/edm_oracle.py) against golden vectors produced by
the reference itself (oracle/make_golden.py) and against the known-answer
values of SURVEY.md 8(c).  CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import edm_oracle as O
from oracle.make_golden import tiny_cfgs, grad_digest


def _ld(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_tables_match_reference(golden_dir):
    """tests/test_unet_builder.py:14-30 of the reference + exact table contents."""
    g = _ld(golden_dir, "tables.npz")
    assert len(g["enc_ch"]) == 15 and len(g["dec_ch"]) == 21 and len(g["skips"]) == 21 and len(g["skip_ch"]) == 21
    d = O.DenoiserCfg(encoder_out_channels=list(g["enc_ch"]), decoder_out_channels=list(g["dec_ch"]),
                      skip_connections=[bool(b) for b in g["skips"]])
    assert d.skip_channels() == [int(v) for v in g["skip_ch"]]


def test_l0_ops(golden_dir):
    g = _ld(golden_dir, "ops.npz")
    x, w, a, b = T(g["x"]), T(g["w"]), T(g["a"]), T(g["b"])
    tol = dict(rtol=1e-5, atol=1e-6)
    assert torch.allclose(O.rms_div(x, [1]), T(g["pixel_norm"]), **tol)
    assert torch.allclose(O.weight_normalize(w), T(g["normalize"]), **tol)
    assert torch.allclose(O.wn_conv(x, w), T(g["conv"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(O.wn_linear(x[:, :, 0, 0], w[:, :, 0, 0]), T(g["linear"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(O.mp_silu(x), T(g["mp_silu"]), **tol)
    assert torch.allclose(O.mp_add(a, b, 0.3), T(g["mp_add03"]), **tol)
    assert torch.allclose(O.mp_add(a, b), T(g["mp_add05"]), **tol)
    # resampling is done by the same ATen ops in the oracle
    assert torch.equal(torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest-exact"), T(g["up"]))
    assert torch.allclose(torch.nn.functional.avg_pool2d(x, 2, 2), T(g["down"]), **tol)


def _tiny(golden_dir):
    g = _ld(golden_dir, "tiny_net.npz")
    ecfg, dcfg = tiny_cfgs()
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(7))
    return g, ecfg, dcfg, P


def test_param_regeneration_is_stable(golden_dir):
    g, _, _, P = _tiny(golden_dir)
    assert list(g["param_keys"]) == sorted(P)
    dig = np.stack([grad_digest(P[k]) for k in sorted(P)])
    np.testing.assert_allclose(dig, g["param_digest"], rtol=1e-6, atol=1e-7)


def test_tiny_net_eval_forward(golden_dir):
    g, ecfg, dcfg, P = _tiny(golden_dir)
    noisy, sigma, labels = T(g["noisy"]), T(g["sigma"]), T(g["labels"])
    four, emb = O.embedding_forward(P, ecfg, sigma, labels)
    assert torch.allclose(four, T(g["eval_fourier"]), rtol=1e-5, atol=1e-5)
    assert torch.allclose(emb, T(g["eval_emb"]), rtol=1e-4, atol=1e-5)
    D = O.denoiser_forward(P, dcfg, noisy, sigma, emb)
    assert torch.allclose(D, T(g["eval_D"]), rtol=1e-4, atol=1e-4)
    Du = O.edm_forward(P, ecfg, dcfg, noisy, sigma, None)
    assert torch.allclose(Du, T(g["eval_D_uncond"]), rtol=1e-4, atol=1e-4)
    # scalar sigma + (B,1) labels, the way the solver calls the model
    Ds = O.edm_forward(P, ecfg, dcfg, noisy, torch.tensor(1.7), labels.view(-1, 1))
    assert torch.allclose(Ds, T(g["eval_D_scalar_sigma"]), rtol=1e-4, atol=1e-4)


def test_tiny_net_bf16_mode_is_as_close_as_reference_autocast(golden_dir):
    """The oracle's bf16 rounding placement must not be worse than the reference's
    own bf16-autocast deviation from fp32 (SURVEY 7 'bf16 parity target')."""
    g, ecfg, dcfg, P = _tiny(golden_dir)
    noisy, sigma, labels = T(g["noisy"]), T(g["sigma"]), T(g["labels"])
    ref32, refbf = T(g["eval_D"]), T(g["eval_D_autocast_bf16"])
    Db = O.edm_forward(P, ecfg, dcfg, noisy, sigma, labels, bf16=True)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert rel(Db, ref32) <= max(2.0 * rel(refbf, ref32), 5e-3)


def test_tiny_net_training_step_grads(golden_dir):
    g, ecfg, dcfg, P = _tiny(golden_dir)
    for k in O.trainable_keys(P):
        P[k].requires_grad_(True)
    loss = O.training_loss(P, ecfg, dcfg, T(g["clean"]), T(g["eps"]), T(g["noise"]), -1.2, 1.2, T(g["labels"]))
    loss.backward()
    assert abs(loss.item() - float(g["train_loss"])) <= 1e-4 * abs(float(g["train_loss"]))
    keys = [str(k) for k in g["grad_keys"]]
    assert keys == sorted(O.trainable_keys(P))
    for i, k in enumerate(keys):
        np.testing.assert_allclose(grad_digest(P[k].grad), g["grad_digest"][i], rtol=2e-3, atol=2e-5, err_msg=k)
        # forced weight normalisation side effect (networks.py:32-34)
        np.testing.assert_allclose(grad_digest(P[k]), g["post_param_digest"][i], rtol=1e-5, atol=1e-6, err_msg=k)
    for k in [n[6:] for n in g.files if n.startswith("grad::")]:
        np.testing.assert_allclose(P[k].grad.numpy(), g["grad::" + k], rtol=2e-3, atol=2e-5, err_msg=k)


def test_blocks(golden_dir):
    g = _ld(golden_dir, "blocks.npz")
    emb = T(g["emb"])
    spec = {"enc_plain": (False, False), "enc_down": (True, False), "enc_attn": (False, True),
            "enc_widen": (False, False)}
    for tag, (down, attn) in spec.items():
        P = {"b." + k.split("::p::")[1]: T(g[k]) for k in g.files if k.startswith(tag + "::p::")}
        y = O.encoder_block(P, "b.", T(g[tag + "::x"]), emb, down, attn, 2, 0.3, 0.0, False)
        assert torch.allclose(y, T(g[tag + "::y"]), rtol=1e-4, atol=1e-4), tag
    spec = {"dec_plain": (False, False), "dec_up": (True, False), "dec_skip_attn": (False, True),
            "dec_skip": (False, False)}
    for tag, (up, attn) in spec.items():
        P = {"b." + k.split("::p::")[1]: T(g[k]) for k in g.files if k.startswith(tag + "::p::")}
        skip = T(g[tag + "::skip"]) if (tag + "::skip") in g.files else None
        y = O.decoder_block(P, "b.", T(g[tag + "::x"]), emb, skip, up, attn, 2, 0.3, 0.0, False)
        assert torch.allclose(y, T(g[tag + "::y"]), rtol=1e-4, atol=1e-4), tag


def _blocks64_case(g, tag, names_module):
    """inputs + seeded parameters of one blocks64.npz case (oracle/make_golden_blocks64.py)"""
    from oracle.make_golden_blocks64 import SPECS, block_inputs, block_params
    spec = {s[0]: s for s in SPECS}[tag]
    _, kind, cin, cout, skip, resample, attn, hw, seed = spec
    P = block_params(names_module, seed)
    assert [str(k) for k in g[tag + "::keys"]] == sorted(P), tag
    from oracle.make_golden import grad_digest
    np.testing.assert_allclose(np.stack([grad_digest(P[k]) for k in sorted(P)]), g[tag + "::digest"], rtol=1e-6, atol=1e-7)
    x, sk = block_inputs(cin, skip, hw, seed)
    return spec, P, x, sk


def test_blocks64_oracle_matches_reference(golden_dir):
    """every block flavour at HIP-supported widths (64 channels, head dim 32): oracle == the reference's modules"""
    from oracle.make_golden_blocks64 import SPECS
    import tinyedm_amd as TA
    g = _ld(golden_dir, "blocks64.npz")
    emb = T(g["emb"])
    for tag, kind, cin, cout, skip, resample, attn, hw, seed in SPECS:
        # the tinyedm_amd module only supplies parameter names and shapes (identical to the reference's state_dict)
        if kind == "enc":
            names = TA.networks.EncoderBlock(cin, cout, 64, resample, attn, num_heads=2)
        else:
            names = TA.networks.DecoderBlock(cin, cout, 64, resample, attn, num_heads=2, skip_channels=skip)
        _, P, x, sk = _blocks64_case(g, tag, names)
        P = {"b." + k: v for k, v in P.items()}
        if kind == "enc":
            y = O.encoder_block(P, "b.", x, emb, resample, attn, 2, 0.3, 0.0, False)
        else:
            y = O.decoder_block(P, "b.", x, emb, sk, resample, attn, 2, 0.3, 0.0, False)
        ref = T(g[tag + "::y"])
        rel = ((y - ref).norm() / ref.norm()).item()
        assert rel <= 1e-4, f"{tag}: oracle vs reference rel {rel:.3e}"


def test_solver_tables_bitwise_and_trajectory(golden_dir):
    g, ecfg, dcfg, P = _tiny(golden_dir)
    s = _ld(golden_dir, "solver.npz")
    assert np.array_equal(O.karras_schedule(18).numpy().view(np.uint32), s["t18"].view(np.uint32))
    assert np.array_equal(O.karras_schedule(32).numpy().view(np.uint32), s["t32"].view(np.uint32))
    t5 = O.karras_schedule(5, 0.01, 20.0, 5.0)
    assert np.array_equal(t5.numpy().view(np.uint32), s["t5"].view(np.uint32))
    # SURVEY 8(c) oracle facts
    assert s["t32"].dtype == np.float32 and s["t32"].shape == (33,)
    assert float(s["t32"][0]) == 79.99998474121094 and float(s["t32"][32]) == 0.0
    model = lambda x, t, lab: O.edm_forward(P, ecfg, dcfg, x, t, lab)
    with torch.no_grad():
        x5 = O.heun_solve(model, T(s["x0"]), t5, T(s["labels"]))
        x18 = O.heun_solve(model, T(s["x0"]), O.karras_schedule(18), None)
    assert torch.allclose(x5, T(s["x_heun5"]), rtol=1e-3, atol=1e-3)
    assert torch.allclose(x18, T(s["x_heun18_uncond"]), rtol=1e-3, atol=1e-3)


def test_known_answers_optimizer_side():
    """SURVEY 8(c) KATs (fp64 restatements of edm.py:212,305-320; ema.py:29-32,273; networks.py:578-581,165)."""
    assert abs(O.sigma_rel_to_gamma(0.13) - 4.603596781479866) < 1e-9
    assert abs(O.sigma_rel_to_gamma(0.1) - 6.937203937601809) < 1e-9
    assert abs(O.sigma_rel_to_gamma(0.05) - 16.972198602303447) < 1e-9
    gam = O.sigma_rel_to_gamma(0.13)
    for step, want in [(0, 0.0), (1, 0.0205659741), (9, 0.5541067915), (99, 0.9452384742), (999, 0.9944092861)]:
        assert abs(O.ema_beta(step, gam) - want) < 1e-9
    for step, want in [(0, 1e-8), (100, 0.500000005), (199, 0.99500000005), (200, 1), (400, 1), (600, 0.70710678),
                       (1000, 0.5)]:
        assert abs(O.lr_lambda(step, 200, 200) - want) < 1e-8
    w = O.loss_weight(torch.tensor([0.5, 80.0, 0.002], dtype=torch.float64), 0.5)
    assert torch.allclose(w, torch.tensor([8.0, 4.00015625, 250004.0], dtype=torch.float64))
    cs, co, ci = O.precond_scalars(torch.tensor([0.5]), 0.5)
    assert abs(cs.item() - 0.5) < 1e-7 and abs(co.item() - 0.35355339) < 1e-7 and abs(ci.item() - 1.41421356) < 1e-6
    assert abs(math.log(0.5) / 4 - (-0.17328680)) < 1e-8


def test_weighted_mse_closed_form():
    """Reference tests/test_weighted_mean_squared_error.py:18-21."""
    g = torch.Generator().manual_seed(0)
    w = torch.randn(8, generator=g).exp()
    p, t = torch.randn(8, 3, 32, 32, generator=g), torch.randn(8, 3, 32, 32, generator=g)
    assert torch.allclose(O.weighted_mse(w, p, t), torch.mean(w.view(-1, 1, 1, 1) * (p - t) ** 2))


def test_adam_matches_torch_optim():
    g = torch.Generator().manual_seed(3)
    th = torch.randn(1000, generator=g)
    p = torch.nn.Parameter(th.clone())
    opt = torch.optim.Adam([p], lr=0.02, betas=(0.9, 0.999))
    m, v = torch.zeros_like(th), torch.zeros_like(th)
    for step in range(1, 6):
        gr = torch.randn(1000, generator=g)
        p.grad = gr.clone()
        opt.step()
        O.adam_step(th, gr, m, v, step, 0.02)
        assert torch.allclose(th, p.data, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["mnist", "cifar10_cond", "default32"])
def test_full_size_configs_eval_forward_matches_reference(golden_dir, name):
    """The oracle on the reference's full-size configurations (MNIST 87 M, CIFAR-10 conditional 35.6 M, default
    ImageNet / latent net 272.9 M parameters; head dims 64/128/144/192) vs outputs the reference itself produced
    (oracle/make_golden_configs.py)."""
    from oracle.make_golden_configs import config_cases

    class _Tables:   # the default-architecture tables, pinned separately by tests/golden/tables.npz
        pass
    t = np.load(os.path.join(golden_dir, "tables.npz"))
    tb = _Tables()
    tb.get_encoder_blocks_types = lambda: [str(s) for s in t["enc_types"]]
    tb.get_decoder_blocks_types = lambda: [str(s) for s in t["dec_types"]]
    tb.get_encoder_out_channels = lambda: [int(v) for v in t["enc_ch"]]
    tb.get_decoder_out_channels = lambda: [int(v) for v in t["dec_ch"]]
    tb.get_skip_connections = lambda: [bool(v) for v in t["skips"]]
    ecfg, dcfg, shape, seed = config_cases(tb)[name]
    g = np.load(os.path.join(golden_dir, "configs.npz"))
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(seed))
    noisy, sigma, labels = T(g[name + "::noisy"]), T(g[name + "::sigma"]), T(g[name + "::labels"])
    assert tuple(noisy.shape) == shape
    with torch.no_grad():
        D = O.edm_forward(P, ecfg, dcfg, noisy, sigma, labels)
    ref = T(g[name + "::D"])
    rel = ((D - ref).norm() / ref.norm()).item()
    assert rel <= 1e-4, f"{name}: oracle vs reference rel {rel:.3e}"


def test_uncertainty_net_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "uncertainty.npz"))
    U = {"u.linear1.weight": T(g["w1"]), "u.linear2.weight": T(g["w2"]), "u.gain": torch.tensor(float(g["gain"]))}
    y = O.uncertainty_forward(U, T(g["x"]))
    assert torch.allclose(y, T(g["y"]), rtol=1e-5, atol=1e-6)
