"""Oracle checks of the kernel instantiations the benchmark actually dispatches: the full-size
layers -- B = 128, 32x32 (512x128-tile v6 kernel), 16x16 (512x64 tile) and 8x8 -- with their fused epilogues, each
compared on images {0, 63, 127} of the batch against an fp64 evaluation of the SAME bf16 operands (reference:
F.conv2d networks.py:37 and the elementwise chain networks.py:253-260 / 319-324 and its autograd), never against
another HIP kernel; and the CIFAR-10 unconditional network at B = 128 against the oracle with bf16 rounding points."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import edm_oracle as O
from parity_log import record

DEV = "cuda"
IMGS = [0, 63, 127]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tinyedm_amd import ops as _ops
    return _ops


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def rt(x):          # bf16 rounding, as a double tensor
    return x.to(torch.bfloat16).double()


def nchw64(x_nhwc):  # NHWC bf16 (any device) -> NCHW fp64 on the CPU
    return x_nhwc.float().cpu().permute(0, 3, 1, 2).double()


def _layer(B, HW, Cin, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, HW, HW, Cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(torch.bfloat16)
    wp = w.permute(2, 3, 0, 1).reshape(9, Cout, Cin).contiguous()
    return x, w, wp


def _conv64(x, w):   # fp64 conv of the chosen images, NCHW
    return F.conv2d(nchw64(x[IMGS]), w.double(), padding=1)


SHAPES = [(128, 32, 256, 256, "v6 512x128"), (128, 32, 512, 256, "v6 512x128"), (128, 16, 256, 256, "v6 512x64"),
          (128, 8, 256, 256, "8x8")]


@pytest.mark.parametrize("B,HW,Cin,Cout,kern", SHAPES)
def test_full_size_conv_plain_and_residual(ops, B, HW, Cin, Cout, kern):
    x, w, wp = _layer(B, HW, Cin, Cout, 1)
    r = torch.randn(B, HW, HW, Cout, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16)
    ref = _conv64(x, w)
    y = ops.conv_igemm(x.to(DEV), wp.to(DEV), 9)
    e = rel(nchw64(y[IMGS]), ref)
    record(f"fullsize/conv3x3[{kern} {Cin}->{Cout} {HW}x{HW}]", e, 4e-3)
    assert e <= 4e-3, e
    y2 = ops.conv_igemm(x.to(DEV), wp.to(DEV), 9, residual=r.to(DEV), alpha=0.7, beta=0.3)
    e2 = rel(nchw64(y2[IMGS]), 0.7 * ref + 0.3 * nchw64(r[IMGS]))
    record(f"fullsize/conv3x3+residual[{kern} {Cin}->{Cout} {HW}x{HW}]", e2, 4e-3)
    assert e2 <= 4e-3, e2


@pytest.mark.parametrize("B,HW,Cin,Cout,kern", SHAPES)
def test_full_size_mod_epilogues_against_the_oracle_chain(ops, B, HW, Cin, Cout, kern):
    """forward: u = conv(x) (bf16), a2 = dropout(mp_silu(u*(lin*gain+1)));  backward of the block's second conv with
    the modulation backward in its epilogue: gr = d a2/d u applied to ga = alpha*conv(dy, wd), glin per sample.
    Reference: the same chain in fp64 torch + autograd (the keep mask is data: the kernel's own Philox mask)."""
    pdrop, seed, sub, step = 0.13, 1234, 7, 3
    x, w, wp = _layer(B, HW, Cin, Cout, 3)
    g = torch.Generator().manual_seed(4)
    lin = torch.randn(B, Cout, generator=g)
    gain = torch.tensor(0.6)
    u, a2 = ops.conv3x3_mod(x.to(DEV), wp.to(DEV), lin.to(DEV), gain.to(DEV), pdrop, seed, sub, step, want_u=True)
    mask = ops.dropout_mask(B * HW * HW * Cout, pdrop, seed, sub, step, DEV).view(B, HW, HW, Cout)
    keep = mask[IMGS].cpu().permute(0, 3, 1, 2).double() / (1.0 - pdrop)
    u_ref = _conv64(x, w)
    e_u = rel(nchw64(u[IMGS]), u_ref)
    m = (lin[IMGS].double() * gain.double() + 1.0).view(len(IMGS), Cout, 1, 1)
    ub = rt(u_ref)                                       # the kernel modulates the bf16-rounded conv output
    a2_ref = O.mp_silu(ub * m) * keep
    e_a = rel(nchw64(a2[IMGS]), a2_ref)
    record(f"fullsize/conv3x3_mod.u[{kern} {Cin}->{Cout} {HW}x{HW}]", e_u, 4e-3)
    record(f"fullsize/conv3x3_mod.a2[{kern} {Cin}->{Cout} {HW}x{HW}]", e_a, 6e-3)
    assert e_u <= 4e-3 and e_a <= 6e-3, (e_u, e_a)

    # ---- backward: second conv of the block is Cout -> Cout; its dgrad feeds the modulation backward
    gy = torch.randn(B, HW, HW, Cout, generator=g).to(torch.bfloat16)
    w2 = (torch.randn(Cout, Cout, 3, 3, generator=g) / math.sqrt(Cout * 9)).to(torch.bfloat16)
    wd = w2.flip(2, 3).permute(2, 3, 1, 0).reshape(9, Cout, Cout).contiguous()     # dgrad pack [8-tap][ci][co]
    alpha = 0.55
    r1 = u                                                # pre-activation saved by the forward
    gr, glin, ggain = ops.conv3x3_modbwd(gy.to(DEV), wd.to(DEV), alpha, r1, lin.to(DEV), gain.to(DEV), pdrop, seed, sub, step)
    ga_ref = alpha * F.conv_transpose2d(nchw64(gy[IMGS]), w2.double(), padding=1)  # dgrad of conv(.., w2)
    ur = nchw64(u[IMGS]).requires_grad_(True)
    linr = lin[IMGS].double().clone().requires_grad_(True)
    a2_fn = O.mp_silu(ur * (linr * gain.double() + 1.0).view(len(IMGS), Cout, 1, 1)) * keep
    a2_fn.backward(rt(ga_ref))                            # the kernel rounds ga to bf16 before the epilogue
    e_gr = rel(nchw64(gr[IMGS]), ur.grad)
    e_gl = rel(glin[IMGS].cpu(), linr.grad)
    record(f"fullsize/conv3x3_modbwd.gr[{kern} {HW}x{HW}]", e_gr, 8e-3)
    record(f"fullsize/conv3x3_modbwd.glin[{kern} {HW}x{HW}]", e_gl, 8e-3)
    assert e_gr <= 8e-3 and e_gl <= 8e-3, (e_gr, e_gl)


@pytest.mark.parametrize("B,HW,Cin,Cout,kern", SHAPES)
def test_full_size_silubwd_epilogue(ops, B, HW, Cin, Cout, kern):
    """dgrad of a decoder block's first conv with the mp_silu backward fused: gx = mp_silu'(xpre)*conv^T(g) + s*extra"""
    g = torch.Generator().manual_seed(6)
    gy = torch.randn(B, HW, HW, Cout, generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(torch.bfloat16)
    wd = w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, Cin, Cout).contiguous()
    xpre = torch.randn(B, HW, HW, Cin, generator=g).to(torch.bfloat16)
    extra = torch.randn(B, HW, HW, Cin, generator=g).to(torch.bfloat16)
    gx = ops.conv3x3_silubwd(gy.to(DEV), wd.to(DEV), xpre.to(DEV), extra.to(DEV), 0.4)
    gref = rt(F.conv_transpose2d(nchw64(gy[IMGS]), w.double(), padding=1))
    xp = nchw64(xpre[IMGS]).requires_grad_(True)
    O.mp_silu(xp).backward(gref)
    ref = xp.grad + 0.4 * nchw64(extra[IMGS])
    e = rel(nchw64(gx[IMGS]), ref)
    record(f"fullsize/conv3x3_silubwd[{kern} {Cout}->{Cin} {HW}x{HW}]", e, 6e-3)
    assert e <= 6e-3, e


def test_cifar10_unconditional_forward_b128_vs_oracle():
    """BASELINE configs[1] as benchmarked: CIFAR-10 unconditional U-Net (35.6 M parameters), batch 128, eval forward;
    images {0, 63, 127} against O.edm_forward with the same bf16 rounding points (on D - c_skip*x)."""
    import tinyedm_amd as T
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ecfg, dcfg = O.cifar10_cfg()
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(1), gains_nonzero=True)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), dcfg.dropout_rate,
                     dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
    den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
    emb, den = emb.to(DEV).eval(), den.to(DEV).eval()
    g = torch.Generator().manual_seed(12)
    B = 128
    clean = 0.5 * torch.randn(B, 3, 32, 32, generator=g)
    sigma = (torch.randn(B, generator=g) * 1.2 - 1.2).exp()
    noisy = clean + sigma.view(-1, 1, 1, 1) * torch.randn(B, 3, 32, 32, generator=g)
    with torch.no_grad():
        _, e = emb(sigma.to(DEV), None)
        D = den(noisy.to(DEV), sigma.to(DEV), e).cpu()
        D_or = O.edm_forward(P, ecfg, dcfg, noisy[IMGS], sigma[IMGS], None, bf16=True)
    c_skip, _, _ = O.precond_scalars(sigma[IMGS], dcfg.sigma_data)
    base = c_skip * noisy[IMGS]
    err = rel(D[IMGS] - base, D_or - base)
    # (limit: profiles/r03_error_growth.json -- two independent bf16 evaluations of this net are ~1e-2 apart)
    record("fullsize/cifar10_uncond_B128_forward_vs_bf16_oracle", err, 2.5e-2)
    assert err <= 2.5e-2, err
    # the same batch through the reference-precision path against the FP32 oracle
    den.set_eval_dtype("f32")
    with torch.no_grad():
        D32 = den(noisy.to(DEV), sigma.to(DEV), e).cpu()
        D_or32 = O.edm_forward(P, ecfg, dcfg, noisy[IMGS], sigma[IMGS], None, bf16=False)
    den.set_eval_dtype("bf16")
    err32 = rel(D32[IMGS] - base, D_or32 - base)
    record("fullsize/cifar10_uncond_B128_forward_f32_path_vs_fp32_oracle", err32, 1e-4)
    assert err32 <= 1e-4, err32


def test_fragment_major_packs_b128():
    """Round 4: at batch 128 the 8x8 layers of the CIFAR-10 net run on k_conv3x3_s with FRAGMENT-MAJOR weight packs
    (csrc/weights.hip; coalesced 1-KiB loads straight into MFMA operand registers).  Same values as the [tap][co][ci] packs:
    the evaluation forward bit for bit, every parameter gradient of a training step up to the order of the fp32 atomics
    both paths share -- and the plan really marks those layers (the 16 convs of the eight 8x8 blocks: 16 forward + 16 dgrad packs)."""
    import tinyedm_amd as T
    from tinyedm_amd import networks as N
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ecfg, dcfg = O.cifar10_cfg()
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(1), gains_nonzero=True)
    g = torch.Generator().manual_seed(12)
    B = 128
    clean = (0.5 * torch.randn(B, 3, 32, 32, generator=g)).to(DEV)
    sigma = (torch.randn(B, generator=g) * 1.2 - 1.2).exp().to(DEV)
    res = {}
    old = N.FRAG_PACKS
    try:
        for mode in (False, True):
            N.FRAG_PACKS = mode
            N._rng_sub_counter[0] = 0
            T.manual_seed(5)
            emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
            den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                             tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                             tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), dcfg.dropout_rate,
                             dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim,
                             dcfg.num_heads)
            emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
            den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
            model = T.EDM(diffuser=T.Diffuser(-1.2, 1.2), embedding=emb, denoiser=den, use_ema=False, use_uncertainty=False,
                          steady_steps=10, rampup_steps=2, scheduler_interval="step", lr=1e-3).to(DEV)
            base = model.configure_optimizers()["optimizer"]
            model.eval()
            with torch.no_grad():
                D = model(clean, sigma, None).clone()
            nfrag = sum(1 for plan in den._plans.values() for (wf, wd, _) in plan.caches
                        for t in (wf, wd) if t is not None and getattr(t, "_edm_frag", False))
            model.train()
            base.zero_grad()
            T.manual_seed(5)
            loss = model.training_step((clean, None), 0)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (D, float(loss), base.arena.grad.clone(), nfrag)
    finally:
        N.FRAG_PACKS = old
    assert res[False][3] == 0 and res[True][3] == 32, (res[False][3], res[True][3])
    assert torch.equal(res[True][0], res[False][0])
    assert abs(res[True][1] - res[False][1]) <= 1e-6 * abs(res[False][1])
    e = rel(res[True][2], res[False][2])
    assert e <= 1e-5, e


def test_captured_solve_sees_new_weights_when_the_plan_is_fragment_major():
    """A captured Heun solve reads the weight packs of the plan of ITS input shape (at batch 128 the one with fragment-major
    packs on the 8x8 layers); the refresh before a replay (solvers.py) is called without a shape and must bring THAT plan up
    to date too: after an in-place change of an 8x8 layer's weight the replay equals the eager solve, not the old one."""
    import tinyedm_amd as T
    from tinyedm_amd import _runtime_env
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if not _runtime_env.GRAPH_REPLAY_SAFE:
        pytest.skip("hipGraph replay needs the safe runtime setting")
    ecfg, dcfg = O.cifar10_cfg()
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(1), gains_nonzero=True)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), dcfg.dropout_rate,
                     dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
    den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
    model = T.EDM(diffuser=T.Diffuser(-1.2, 1.2), embedding=emb, denoiser=den, use_ema=False, use_uncertainty=False,
                  steady_steps=10, rampup_steps=2, scheduler_interval="step", lr=1e-3).to(DEV).eval()
    x0 = torch.randn(128, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(DEV)
    solver = T.DeterministicSolver(num_steps=2)
    with torch.no_grad():
        x1 = solver.solve(model, x0, None, graph=True)
        assert any(getattr(t, "_edm_frag", False) for plan in den._plans.values() for c in plan.caches for t in c[:2] if t is not None)
        w = den.encoder_blocks[6].conv_3x3_1.weight          # an 8x8 layer: fragment-major packs at this batch
        w.mul_(torch.linspace(0.5, 1.5, w.shape[1], device=DEV).view(1, -1, 1, 1))
        x2 = solver.solve(model, x0, None, graph=True)
        x2e = solver.solve(model, x0, None)
    assert torch.equal(x2, x2e)
    assert not torch.equal(x2, x1)


def _cifar_model(T, seed=1):
    ecfg, dcfg = O.cifar10_cfg()
    P = O.init_params(ecfg, dcfg, torch.Generator().manual_seed(seed), gains_nonzero=True)
    emb = T.Embedding(ecfg.fourier_dim, ecfg.embedding_dim, ecfg.num_classes, ecfg.add_factor)
    den = T.Denoiser(dcfg.in_channels, dcfg.out_channels, tuple(dcfg.encoder_block_types),
                     tuple(dcfg.decoder_block_types), tuple(dcfg.encoder_out_channels),
                     tuple(dcfg.decoder_out_channels), tuple(dcfg.skip_connections), dcfg.dropout_rate,
                     dcfg.sigma_data, dcfg.encoder_add_factor, dcfg.decoder_add_factor, dcfg.embedding_dim, dcfg.num_heads)
    emb.load_state_dict({k[len("embedding."):]: v for k, v in P.items() if k.startswith("embedding.")})
    den.load_state_dict({k[len("denoiser."):]: v for k, v in P.items() if k.startswith("denoiser.")})
    model = T.EDM(diffuser=T.Diffuser(-1.2, 1.2), embedding=emb, denoiser=den, use_ema=False, use_uncertainty=False,
                  steady_steps=10, rampup_steps=2, scheduler_interval="step", lr=1e-3).to(DEV)
    return model, den


def test_captured_f32x3_solve_sees_new_weights():
    """the split-bf16 weight packs of the "f32x3" evaluation are persistent plan buffers rewritten in
    place by the refresh before a replay -- after an in-place weight change (and after an eager f32x3 forward in between,
    which used to re-allocate the pack under the live graph) the replayed solve equals the eager one."""
    import tinyedm_amd as T
    from tinyedm_amd import _runtime_env
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if not _runtime_env.GRAPH_REPLAY_SAFE:
        pytest.skip("hipGraph replay needs the safe runtime setting")
    model, den = _cifar_model(T)
    model.eval()
    den.set_eval_dtype("f32x3")
    x0 = torch.randn(16, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(DEV)
    solver = T.DeterministicSolver(num_steps=2)
    with torch.no_grad():
        x1 = solver.solve(model, x0, None, graph=True)
        conv = den.encoder_blocks[2].conv_3x3_1
        addr = conv._split_pack.data_ptr()
        w = conv.weight
        w.mul_(torch.linspace(0.5, 1.5, w.shape[1], device=DEV).view(1, -1, 1, 1))
        w2 = den.decoder_blocks[2].attention.qkv_conv.weight
        w2.mul_(torch.linspace(1.5, 0.5, w2.shape[1], device=DEV).view(1, -1, 1, 1))
        x2e = solver.solve(model, x0, None)                 # an eager f32x3 pass between capture and replay
        assert conv._split_pack.data_ptr() == addr          # rewritten in place, not re-allocated
        x2 = solver.solve(model, x0, None, graph=True)
    assert torch.equal(x2, x2e)
    assert not torch.equal(x2, x1)
    den.set_eval_dtype("bf16")


def test_eval_dtype_toggle_leaves_the_training_plan_alone():
    """a sampling callback that switches the evaluation precision for a while must not change the
    training plan's key (set_eval_dtype used to append "hat" to every module's _want for good: a second full plan, and fp32
    effective weights written by every later training step), and repeated toggles / shapes keep the number of plans bounded."""
    import tinyedm_amd as T
    from tinyedm_amd.callbacks import _eval_dtype
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    model, den = _cifar_model(T)
    g = torch.Generator().manual_seed(5)
    x = (0.5 * torch.randn(8, 3, 32, 32, generator=g)).to(DEV)
    sigma = torch.ones(8, device=DEV)
    model.train()
    model.training_step((x, None), 0).backward()
    train_keys = set(den._plans)
    wants = {m: tuple(m._want) for m in den.modules() if hasattr(m, "_want")}
    for _ in range(3):
        for dt in ("f32x3", "f32"):
            model.eval()
            with _eval_dtype(model, dt), torch.no_grad():
                model(x, sigma, None)
            model.train()
            model.training_step((x, None), 0).backward()
    assert all(tuple(m._want) == w for m, w in wants.items())
    assert train_keys <= set(den._plans)                         # the training plan is still there, under the same key
    (tk,) = train_keys
    assert all("hat" not in w or "hat" in m._want for m, w in zip(den._plans[tk].mods, den._plans[tk].wants))
    assert len(den._plans) == 3                                  # training, f32x3 evaluation, f32 evaluation
    model.eval()
    with torch.no_grad():
        for b in range(1, 12):                                   # many shapes: the plan count stays bounded
            model(x[:1].expand(b, -1, -1, -1).contiguous(), sigma[:1].expand(b).contiguous(), None)
    assert len(den._plans) <= den.MAX_PLANS + 1
    torch.cuda.synchronize()


def test_standalone_module_call_after_a_fragment_major_forward():
    """a Denoiser forward at batch 128 leaves FRAGMENT-MAJOR packs in the 8x8 convs' caches (eval mode keeps
    caches); the reference-style standalone module call on another shape must still work (plain packs) and must not disturb
    the plan's buffers."""
    import tinyedm_amd as T
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    model, den = _cifar_model(T)
    model.eval()
    g = torch.Generator().manual_seed(5)
    x = (0.5 * torch.randn(128, 3, 32, 32, generator=g)).to(DEV)
    sigma = torch.ones(128, device=DEV)
    with torch.no_grad():
        D1 = model(x, sigma, None).clone()
        conv = den.encoder_blocks[6].conv_3x3_1
        assert getattr(conv._cache[0], "_edm_frag", False)
        xin = torch.randn(2, conv.in_channels, 8, 8, generator=g).to(DEV)
        y = conv(xin)                                            # NCHW module API, a shape k_conv3x3_s does not run
        ref = O.wn_conv(xin.cpu().float(), conv.weight.detach().cpu().float())
        assert rel(y.cpu().float(), ref) < 1e-2
        D2 = model(x, sigma, None)
    assert torch.equal(D1, D2)


def test_resample_fusions_leave_the_cifar10_step_unchanged():
    """FUSE_RESAMPLE (round 6; the CIFAR-10 net has two EncD blocks without a 1x1 conv and two DecU blocks) and the
    producer-written mp_silu of the decoder blocks without a skip (Denoiser._silu_dest, rides on FUSE_CAT): the evaluation
    forward bit for bit, the training loss and every parameter gradient up to the order of the fp32 atomics both
    paths share -- against the sequences with the standalone pool / upsample / mp_silu / concat kernels."""
    import tinyedm_amd as T
    from tinyedm_amd import networks as N
    from tinyedm_amd.ema import FusedAdam
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    g = torch.Generator().manual_seed(12)
    x = (0.5 * torch.randn(8, 3, 32, 32, generator=g)).to(DEV)
    sigma = torch.randn(8, generator=g).exp().to(DEV)
    res = {}
    old = (N.FUSE_RESAMPLE, N.FUSE_CAT, N.SG_MULTI, N.SG_BWD_MULTI, N.SG_HALVES)
    # (third flag, SG_MULTI: every decoder gate of a pass from one launch behind the encoder; fourth, SG_BWD_MULTI: their
    # backward deferred to the last of them, placeholders for the skip gradients -- same kernel bodies, same values)
    # (fifth, SG_HALVES: that launch also writes the gated-skip halves of the decoder's cat / mp_silu(cat) buffers)
    modes = [(False, True, False, False, False), (True, True, True, True, True), (True, False, True, True, True),
             (True, True, False, False, False), (True, True, True, False, True), (True, True, True, True, False)]
    try:
        for mode in modes:
            N.FUSE_RESAMPLE, N.FUSE_CAT, N.SG_MULTI, N.SG_BWD_MULTI, N.SG_HALVES = mode
            N._rng_sub_counter[0] = 0           # the same dropout sub-streams for both models
            model, den = _cifar_model(T, seed=4)
            opt = FusedAdam(model.parameters(), lr=1e-3)
            opt.zero_grad()
            model.eval()
            with torch.no_grad():
                D = model(x, sigma, None).clone()
            model.train()
            T.manual_seed(5)
            loss = model.training_step((x, None), 0)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (D, float(loss), opt.arena.grad.clone())
    finally:
        N.FUSE_RESAMPLE, N.FUSE_CAT, N.SG_MULTI, N.SG_BWD_MULTI, N.SG_HALVES = old
    ref = res[modes[0]]
    for mode in modes[1:]:
        assert torch.equal(res[mode][0], ref[0]), mode
        assert abs(res[mode][1] - ref[1]) <= 1e-6 * abs(ref[1]), mode      # (the loss is a sum of fp32 atomics)
        e = rel(res[mode][2], ref[2])
        assert e <= 1e-5, (mode, e)
