"""Per-kernel parity: every C-ABI HIP kernel against the CPU oracle / plain fp32 torch math on the
same seeded inputs.  Tolerances: kernels take bf16 operands and accumulate in fp32, so outputs
are compared with the fp64 result of the SAME bf16-rounded operands; the only error left is the
final bf16 rounding of the output (2^-9 relative) plus summation order."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import edm_oracle as O


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tinyedm_amd import ops as _ops
    return _ops


DEV = "cuda"


def q(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc(x):  # NCHW fp32 (cpu) -> NHWC bf16 (gpu)
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def nchw(y):  # NHWC bf16 (gpu) -> NCHW fp64 (cpu)
    return y.float().cpu().permute(0, 3, 1, 2).double()


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def close_bf16(y, ref, l2=4e-3, mx=1.5e-2):
    ref = ref.double()
    y = y.double()
    assert torch.isfinite(y).all()
    assert rel(y, ref) <= l2, f"rel L2 {rel(y, ref):.3e}"
    assert (y - ref).abs().max().item() <= mx * ref.abs().max().item() + 1e-6, \
        f"max err {(y - ref).abs().max().item():.3e} vs scale {ref.abs().max().item():.3e}"


def pack_fwd(w):  # (O,I,k,k) -> (taps,O,I) bf16
    O_, I_, k, _ = w.shape
    return w.permute(2, 3, 0, 1).reshape(k * k, O_, I_).contiguous().to(torch.bfloat16).to(DEV)


CONV_SHAPES = [
    # B, H, W, Cin, Cout
    (2, 8, 8, 64, 64),
    (3, 16, 16, 128, 192),
    (1, 32, 32, 64, 128),
    (5, 7, 7, 64, 64),
    (2, 14, 14, 32, 64),
    (2, 8, 8, 512, 256),
    (1, 64, 64, 64, 64),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", CONV_SHAPES)
@pytest.mark.parametrize("k", [3, 1])
def test_conv_igemm_forward(ops, B, H, W, Cin, Cout, k):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin + k)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    ref = F.conv2d(x.double(), w.double(), padding=k // 2)
    y = ops.conv_igemm(nhwc(x), pack_fwd(w), k * k)
    close_bf16(nchw(y), ref)


def test_conv_igemm_residual_epilogue(ops):
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 16, 16, 128, 128
    x = q(torch.randn(B, Cin, H, W, generator=g))
    r = q(torch.randn(B, Cout, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    ref = 0.7 * F.conv2d(x.double(), w.double(), padding=1) + 0.3 * r.double()
    y = ops.conv_igemm(nhwc(x), pack_fwd(w), 9, residual=nhwc(r), alpha=0.7, beta=0.3)
    close_bf16(nchw(y), ref)


@pytest.mark.parametrize("B,H,W,Cin,Cout", CONV_SHAPES[:6])
@pytest.mark.parametrize("k", [3, 1])
def test_weight_prep_dgrad_and_wgrad(ops, B, H, W, Cin, Cout, k):
    """weight_prep packs + in-place normalisation, conv dgrad through the flipped pack, and
    wgrad slabs + finish (projection through the weight normalisation) vs oracle autograd."""
    g = torch.Generator().manual_seed(B * 77 + H + Cin + k)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w0 = torch.randn(Cout, Cin, k, k, generator=g) * 1.7
    gy = q(torch.randn(B, Cout, H, W, generator=g))
    # oracle: training-mode forward side effect then differentiable effective weight
    wm = O.weight_normalize(w0).clone().requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xg, O.q_bf16(O.effective_weight(wm)), padding=k // 2)
    y_ref.backward(gy)

    wd = w0.clone().to(DEV)
    wf, wdg, wh = ops.weight_prep(wd, k * k, want_hat=True, normalize_inplace=True)
    assert torch.allclose(wd.cpu(), wm.detach(), rtol=1e-5, atol=1e-6)          # in-place normalisation
    assert torch.allclose(wh.cpu().view_as(w0), O.effective_weight(wm).detach(), rtol=1e-5, atol=1e-7)
    # bf16 pack: identical up to one bf16 ulp where the fp32 value sits on a rounding boundary
    exp_pack = pack_fwd(O.effective_weight(wm).detach()).cpu().float()
    assert torch.allclose(wf.cpu().float(), exp_pack, rtol=2 ** -7, atol=0)
    assert (wf.cpu().float() != exp_pack).float().mean().item() < 1e-3

    y = ops.conv_igemm(nhwc(x), wf, k * k)
    close_bf16(nchw(y), y_ref.detach())
    gx = ops.conv_igemm(nhwc(gy), wdg, k * k)                                      # dgrad
    close_bf16(nchw(gx), xg.grad)

    slabs = ops.conv_wgrad(nhwc(x), nhwc(gy), k * k)
    gw = ops.wgrad_finish(slabs, wd, k * k, Cin)
    assert rel(gw.cpu(), wm.grad) <= 2e-3, f"wgrad rel {rel(gw.cpu(), wm.grad):.3e}"
    # raw (unprojected) wgrad sanity: sum of slabs equals dL/dw_hat in packed order
    what = O.effective_weight(wm).detach().requires_grad_(True)
    F.conv2d(x, what, padding=k // 2).backward(gy)
    raw = slabs.sum(0).cpu()                                                       # (taps, O, I)
    raw_ref = what.grad.permute(2, 3, 0, 1).reshape(k * k, Cout, Cin)
    assert rel(raw, raw_ref) <= 1e-3, f"raw wgrad rel {rel(raw, raw_ref):.3e}"


def test_weight_prep_padding_and_perm(ops):
    g = torch.Generator().manual_seed(11)
    w0 = torch.randn(64, 4, 3, 3, generator=g)
    perm = torch.randperm(64, generator=g).to(torch.int32)
    wd = w0.clone().to(DEV)
    wf, _, wh = ops.weight_prep(wd, 9, Ipad=32, want_dgrad=False, want_hat=True, perm=perm.to(DEV))
    eff = O.effective_weight(w0)
    assert torch.allclose(wd.cpu(), w0)                                           # eval mode: master untouched
    assert torch.allclose(wh.cpu().view_as(w0), eff, rtol=1e-5, atol=1e-7)
    exp = torch.zeros(9, 64, 32)
    exp[:, :, :4] = eff[perm.long()].permute(2, 3, 0, 1).reshape(9, 64, 4)
    assert torch.equal(wf.cpu(), exp.to(torch.bfloat16))


def test_pixelnorm_silu(ops):
    g = torch.Generator().manual_seed(3)
    for C in (64, 192, 256, 512, 768):
        x = q(torch.randn(2, C, 5, 6, generator=g) * 2)
        xr = x.clone().requires_grad_(True)
        xn_ref = q(O.rms_div(xr, [1]))
        a_ref = O.mp_silu(xn_ref)
        xn, a, d = ops.pixelnorm_silu_fwd(nhwc(x))
        close_bf16(nchw(xn), xn_ref.detach())
        close_bf16(nchw(a), a_ref.detach())
        gxn, ga = q(torch.randn(x.shape, generator=g)), q(torch.randn(x.shape, generator=g))
        (0.8 * (xn_ref * gxn).sum() + (a_ref * ga).sum()).backward()
        gx = ops.pixelnorm_silu_bwd(xn, d, nhwc(gxn), 0.8, nhwc(ga))
        close_bf16(nchw(gx), xr.grad, l2=8e-3, mx=3e-2)


def test_silu_axpby_resample(ops):
    g = torch.Generator().manual_seed(4)
    x = q(torch.randn(2, 64, 8, 8, generator=g))
    y = q(torch.randn(2, 64, 8, 8, generator=g))
    xr = x.clone().requires_grad_(True)
    a_ref = O.mp_silu(xr)
    a_ref.backward(y)
    close_bf16(nchw(ops.silu_fwd(nhwc(x))), a_ref.detach())
    close_bf16(nchw(ops.silu_bwd(nhwc(x), nhwc(y))), xr.grad)
    close_bf16(nchw(ops.silu_bwd(nhwc(x), nhwc(y), nhwc(x), 0.5)), xr.grad + 0.5 * x)
    t = 0.3
    c = math.sqrt((1 - t) ** 2 + t ** 2)
    close_bf16(nchw(ops.axpby(nhwc(x), (1 - t) / c, nhwc(y), t / c)), O.mp_add(x, y, t))
    close_bf16(nchw(ops.pool2(nhwc(x))), F.avg_pool2d(x, 2, 2))
    close_bf16(nchw(ops.up2(nhwc(x))), F.interpolate(x, scale_factor=2, mode="nearest-exact"))


@pytest.mark.parametrize("pdrop", [0.0, 0.13])
def test_mod_silu_dropout(ops, pdrop):
    g = torch.Generator().manual_seed(6)
    B, C, H, W = 3, 128, 16, 16
    r = q(torch.randn(B, C, H, W, generator=g))
    lin = torch.randn(B, C, generator=g) * 0.3
    gain = torch.tensor(0.9)
    ga = q(torch.randn(B, C, H, W, generator=g))
    seed, sub, step = 0x1234567812345678, 17, 5
    rd = nhwc(r)
    a = ops.mod_silu_drop_fwd(rd, lin.to(DEV), gain.to(DEV), pdrop, seed, sub, step)
    mask = ops.dropout_mask(rd.numel(), pdrop, seed, sub, step, DEV).view(B, H, W, C).permute(0, 3, 1, 2).float().cpu()
    if pdrop > 0:
        keep = mask.mean().item()
        assert abs(keep - (1 - pdrop)) < 0.01, keep                       # keep-rate (RNG parity is distributional)
    rr, ll, gg = r.clone().requires_grad_(True), lin.clone().requires_grad_(True), gain.clone().requires_grad_(True)
    ref = O.mp_silu(rr * (ll * gg + 1)[:, :, None, None]) * mask / (1 - pdrop)
    ref.backward(ga)
    close_bf16(nchw(a), ref.detach())
    gr, glin, ggain = ops.mod_silu_drop_bwd(rd, lin.to(DEV), gain.to(DEV), nhwc(ga), pdrop, seed, sub, step)
    close_bf16(nchw(gr), rr.grad)
    assert rel(glin.cpu(), ll.grad) < 2e-3
    assert abs(ggain.item() - gg.grad.item()) <= 2e-3 * abs(gg.grad.item()) + 1e-3


def test_scalelong_concat(ops):
    g = torch.Generator().manual_seed(8)
    B, Ci, Cs, H, W = 2, 64, 128, 8, 8
    P = {"l.layer1.weight": torch.randn(Cs // 16, Cs + 1, 1, 1, generator=g),
         "l.layer2.weight": torch.randn(Cs, Cs // 16, 1, 1, generator=g)}
    inp = q(torch.randn(B, Ci, H, W, generator=g))
    skip = q(torch.randn(B, Cs, H, W, generator=g))
    for v in P.values():
        v.requires_grad_(True)
    sk = skip.clone().requires_grad_(True)
    ii = inp.clone().requires_grad_(True)
    gate_ref = O.scale_long_gate(P, "l.", sk)
    cat_ref = torch.cat((ii, q(sk * gate_ref)), dim=1)
    gcat = q(torch.randn(cat_ref.shape, generator=g))
    cat_ref.backward(gcat)

    w1h = O.effective_weight(P["l.layer1.weight"].detach()).view(Cs // 16, Cs + 1).to(DEV)
    w2h = O.effective_weight(P["l.layer2.weight"].detach()).view(Cs, Cs // 16).to(DEV)
    skd, ind = nhwc(skip), nhwc(inp)
    mean = ops.reduce_hw(skd, scale=1.0 / (H * W))
    assert rel(mean.cpu(), skip.mean(dim=(2, 3))) < 1e-4
    gate, z1 = ops.scalelong_fwd(mean, w1h, w2h)
    assert rel(gate.cpu(), gate_ref.detach().view(B, Cs)) < 1e-4
    cat, sil = ops.concat_gate_fwd(ind, skd, gate, True)
    close_bf16(nchw(cat), cat_ref.detach())
    close_bf16(nchw(sil), O.mp_silu(cat_ref.detach()))
    # backward
    gcd = nhwc(gcat)
    ggate = ops.reduce_hw(gcd, C=Cs, c_off=Ci, y=skd)
    gmean, gw1h, gw2h = ops.scalelong_bwd(mean, w1h, w2h, gate, z1, ggate)
    ginp, gskip = ops.concat_gate_bwd(gcd, gate, gmean, Ci)
    close_bf16(nchw(ginp), ii.grad)
    close_bf16(nchw(gskip), sk.grad, l2=8e-3, mx=3e-2)
    # weight grads: project dW_hat through the normalisation with wgrad_finish (S=1,taps=1)
    for name, gwh in (("l.layer1.weight", gw1h), ("l.layer2.weight", gw2h)):
        wm = P[name].detach().view(gwh.shape).contiguous().to(DEV)
        gw = ops.wgrad_finish(gwh.view(1, 1, *gwh.shape), wm, 1, gwh.shape[1])
        assert rel(gw.cpu(), P[name].grad.view(gwh.shape)) < 5e-3, name


def test_precond_and_conv_out(ops):
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 3, 128, 8, 8
    noisy = torch.randn(B, 3, H, W, generator=g)
    sigma = torch.randn(B, generator=g).exp()
    xin = ops.precond_in(noisy.to(DEV), sigma.to(DEV), 0.5, 32)
    c_skip, c_out, c_in = O.precond_scalars(sigma, 0.5)
    exp = torch.zeros(B, 32, H, W)
    exp[:, :3] = c_in * noisy
    exp[:, 3] = 1
    close_bf16(nchw(xin), q(exp))
    x = q(torch.randn(B, C, H, W, generator=g))
    w = torch.randn(3, C, 1, 1, generator=g)
    gain = torch.tensor(0.7)
    xr, wr, gr = x.clone().requires_grad_(True), w.clone().requires_grad_(True), gain.clone().requires_grad_(True)
    what = O.effective_weight(wr)
    D_ref = F.conv2d(xr, what) * gr * c_out + noisy * c_skip
    dD = torch.randn(B, 3, H, W, generator=g)
    D_ref.backward(dD)
    whd = what.detach().view(3, C).contiguous().to(DEV)
    D, Fraw = ops.conv_out_fwd(nhwc(x), whd, gain.to(DEV), noisy.to(DEV), sigma.to(DEV), 0.5)
    assert rel(D.cpu(), D_ref.detach()) < 1e-5
    gx, gwh, gg = ops.conv_out_bwd(nhwc(x), whd, gain.to(DEV), Fraw, dD.to(DEV), sigma.to(DEV), 0.5)
    close_bf16(nchw(gx), xr.grad)
    assert abs(gg.item() - gr.grad.item()) <= 1e-4 * abs(gr.grad.item()) + 1e-5
    gw = ops.wgrad_finish(gwh.view(1, 1, 3, C), w.view(3, C).contiguous().to(DEV), 1, C)
    assert rel(gw.cpu(), wr.grad.view(3, C)) < 1e-4
    # scalar sigma (solver call pattern, solvers.py:48)
    s0 = torch.tensor([1.7])
    D0, _ = ops.conv_out_fwd(nhwc(x), whd, gain.to(DEV), noisy.to(DEV), s0.to(DEV), 0.5, want_fraw=False)
    cs0, co0, _ = O.precond_scalars(s0, 0.5)
    assert rel(D0.cpu(), (F.conv2d(x, what.detach()) * gain * co0 + noisy * cs0)) < 1e-5
