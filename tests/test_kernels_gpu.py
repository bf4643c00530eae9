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
    import os
    from parity_log import record
    ref = ref.double()
    y = y.double()
    assert torch.isfinite(y).all()
    record("kernels/" + os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0], rel(y, ref), l2)
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
    # small-map kernel (conv_igemm5.hip: Cin % 256 == 0, W <= 16): ragged tiles, partial channel tiles, 2 and 4 rounds
    (16, 8, 8, 256, 256),
    (3, 16, 16, 256, 192),
    (5, 7, 7, 256, 72),
    (2, 4, 4, 512, 128),
    (1, 3, 16, 1024, 64),
]


@pytest.fixture(params=[0, 1, 2, 5, 6],
                ids=["igemm-auto", "igemm-v1", "igemm-v2", "igemm-s", "igemm-v6"])
def igemm_version(request, ops):
    """Every generation of the implicit-GEMM kernel must pass the same parity tests (the per-shape dispatcher
    picks v2 only for full-size layers, so they are forced here on the small test shapes)."""
    old = ops.IGEMM_VERSION
    ops.IGEMM_VERSION = request.param
    yield request.param
    ops.IGEMM_VERSION = old


@pytest.fixture(params=[False, True], ids=["wgrad-v2-all", "wgrad-v2+1x1"])
def wgrad_version(request, ops):
    """the rolling-window kernel for every layer (its 1x1 form), or -- the default -- paired with the dedicated 1x1 kernel
    (generation 1, the register-staged kernel, was retired in round 6)"""
    old1 = ops.WGRAD_1X1
    ops.WGRAD_1X1 = request.param
    yield request.param
    ops.WGRAD_1X1 = old1


@pytest.mark.parametrize("B,H,W,Cin,Cout", CONV_SHAPES)
@pytest.mark.parametrize("k", [3, 1])
def test_conv_igemm_forward(ops, igemm_version, B, H, W, Cin, Cout, k):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin + k)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    ref = F.conv2d(x.double(), w.double(), padding=k // 2)
    y = ops.conv_igemm(nhwc(x), pack_fwd(w), k * k)
    close_bf16(nchw(y), ref)


@pytest.mark.parametrize("B,H,W,Cin,Cout,imgs", [
    (4, 32, 32, 64, 128, None),        # images aligned to tiles, W = 32: static border registers, 64-channel tile
    (3, 16, 32, 128, 64, None),        # one tile per image: zero halo above AND below
    (2, 64, 64, 64, 64, None),         # W = 64: 6-slot slab, four blocks per image row
    (5, 16, 16, 64, 192, None),        # 16x16 images, two per tile (ragged last tile), per-wave top / bottom registers
    (1, 48, 16, 64, 64, None),         # W = 16 but H != 16: the general per-lane validity bits
    (2, 24, 48, 64, 64, None),         # W = 48 (multiple of 16, images not aligned to tiles): general path
    (32, 64, 64, 64, 256, [0, 17, 31]),  # W = 64 with the 128-channel tile (512 x 128 workgroups)
])
def test_conv3x3_v6_border_paths(ops, B, H, W, Cin, Cout, imgs):
    """k_conv3x3_v6 (16x16x32 MFMA form) handles image borders four different ways depending on the shape; every one
    against an fp64 convolution of the same bf16 operands."""
    g = torch.Generator().manual_seed(H * 100 + W + Cout)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    old = ops.IGEMM_VERSION
    ops.IGEMM_VERSION = 6
    try:
        assert ops._igemm_entry(B * H * W, W, Cout, 9, Cin) == "edm_conv_igemm_v6"
        y = ops.conv_igemm(nhwc(x), pack_fwd(w), 9)
    finally:
        ops.IGEMM_VERSION = old
    sel = list(range(B)) if imgs is None else imgs
    ref = F.conv2d(x[sel].double(), w.double(), padding=1)
    close_bf16(nchw(y)[sel], ref)


def test_conv_full_size_layer_properties(ops):
    """BASELINE-size layer (B=128, 32x32, 256->256): too big for an fp64 CPU reference in a test, so check
    size-independent properties of the kernels the dispatcher picks there: linearity of conv in its input,
    agreement between kernel generations, and <dY, conv(X)> == <wgrad(X,dY), W> (adjoint identity)."""
    g = torch.Generator().manual_seed(77)
    B, H, W, C = 128, 32, 32, 256
    x1 = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    x2 = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    wp = (torch.randn(9, C, C, generator=g) / 48).to(torch.bfloat16).to(DEV)
    old = ops.IGEMM_VERSION
    try:
        ys = {}
        for v in (1, 2, 6):
            ops.IGEMM_VERSION = v
            ys[v] = ops.conv_igemm(x1, wp, 9).float()
        ops.IGEMM_VERSION = 0
        y_auto = ops.conv_igemm(x1, wp, 9).float()
    finally:
        ops.IGEMM_VERSION = old
    for v in (2, 6):
        assert rel(ys[v], ys[1]) < 3e-3, (v, rel(ys[v], ys[1]))
    assert torch.equal(y_auto, ys[6])                       # the dispatcher picks the static-schedule kernel here
    y2 = ops.conv_igemm(x2, wp, 9).float()
    y12 = ops.conv_igemm((x1.float() + x2.float()).to(torch.bfloat16), wp, 9).float()
    assert rel(y12, ys[6] + y2) < 1e-2                     # linearity (bf16 rounding of the summed input)
    dy = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    lhs = (dy.float() * ys[6]).sum().item()
    slabs = ops.conv_wgrad(x1, dy, 9)
    rhs = (slabs.sum(0) * wp.float()).sum().item()
    assert abs(lhs - rhs) <= 2e-2 * abs(lhs) + 50.0, (lhs, rhs)


def test_conv_igemm_residual_epilogue(ops, igemm_version):
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
def test_weight_prep_dgrad_and_wgrad(ops, wgrad_version, B, H, W, Cin, Cout, k):
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


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 5, 7, 160, 96), (2, 16, 16, 256, 768), (5, 8, 8, 512, 256),
                                             (1, 3, 3, 32, 32), (16, 16, 16, 128, 320)])
def test_conv_wgrad_1x1_ragged(ops, B, H, W, Cin, Cout):
    """Dedicated 1x1 weight-gradient kernel: partial 256x128 tiles, pixel counts that are not a multiple of the
    64-row stage, empty trailing splits; checked against an fp64 GEMM of the same bf16 operands."""
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    gy = q(torch.randn(B, Cout, H, W, generator=g))
    assert ops.WGRAD_1X1
    slabs = ops.conv_wgrad(nhwc(x), nhwc(gy), 1)
    assert slabs.shape[1:] == (1, Cout, Cin)
    xm = x.permute(0, 2, 3, 1).reshape(-1, Cin).double()
    gm = gy.permute(0, 2, 3, 1).reshape(-1, Cout).double()
    ref = gm.t() @ xm
    got = slabs.double().sum(0)[0].cpu()
    assert rel(got, ref) <= 1e-5, f"1x1 wgrad rel {rel(got, ref):.3e}"
    # the two kernels agree on the same operands (different split counts, same sum)
    ops.WGRAD_1X1 = False
    try:
        ref1 = ops.conv_wgrad(nhwc(x), nhwc(gy), 1).double().sum(0)[0].cpu()
    finally:
        ops.WGRAD_1X1 = True
    assert rel(got, ref1) <= 1e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout,pdrop", [(2, 8, 8, 64, 128, 0.0), (3, 16, 16, 128, 64, 0.25), (1, 5, 7, 64, 72, 0.1),
                                                   (128, 32, 32, 256, 256, 0.13),
                                                   (512, 8, 8, 64, 256, 0.1)])    # W = 8 through k_conv3x3_v6's validity-bit path
def test_conv3x3_mod_epilogue_is_bit_identical_to_separate_kernels(ops, B, H, W, Cin, Cout, pdrop):
    """Fused modulation epilogue (edm_conv3x3_mod) == conv_igemm followed by mod_silu_drop_fwd, bit for bit (same bf16
    rounding of u, same Philox counters), on a small-tile shape, ragged shapes and the full-size 32x32 layer (v6)."""
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = nhwc(q(torch.randn(B, Cin, H, W, generator=g)))
    wp = pack_fwd(q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)))
    lin_all = torch.randn(B, Cout + 40, generator=g).to(DEV)
    lin = lin_all[:, 8:8 + Cout]                                     # strided rows, like the batched embed output
    gain = torch.tensor(0.7, device=DEV)
    u_ref = ops.conv_igemm(x, wp, 9)
    a_ref = ops.mod_silu_drop_fwd(u_ref, lin, gain, pdrop, 1234, 5, 6)
    u, a2 = ops.conv3x3_mod(x, wp, lin, gain, pdrop, 1234, 5, 6)
    assert torch.equal(u, u_ref) and torch.equal(a2, a_ref)
    none_u, a3 = ops.conv3x3_mod(x, wp, lin, gain, pdrop, 1234, 5, 6, want_u=False)
    assert none_u is None and torch.equal(a3, a_ref)
    if pdrop > 0:
        assert 0.5 * pdrop < (a2 == 0).float().mean().item() < 2.0 * pdrop + 0.01


@pytest.mark.parametrize("B,H,W,C1,C2,pdrop", [(2, 8, 8, 64, 128, 0.0), (3, 16, 16, 128, 64, 0.25), (2, 4, 8, 64, 72, 0.1),
                                                 (128, 32, 32, 256, 256, 0.13),
                                                 (512, 8, 8, 64, 256, 0.1)])      # W = 8 through k_conv3x3_v6 (validity bits)
def test_conv3x3_modbwd_epilogue_matches_separate_kernels(ops, B, H, W, C1, C2, pdrop):
    """Fused modulation backward (edm_conv3x3_modbwd + edm_mod_finish) == conv_igemm (dgrad) then mod_silu_drop_bwd:
    gr bit for bit, glin / ggain up to fp32 summation order (both paths accumulate with atomics)."""
    g = torch.Generator().manual_seed(B + C1 + C2)
    gout = nhwc(q(torch.randn(B, C1, H, W, generator=g)))
    wd = pack_fwd(q(torch.randn(C2, C1, 3, 3, generator=g) / math.sqrt(C1 * 9)))      # any [9][C2][C1] pack
    r1 = nhwc(q(torch.randn(B, C2, H, W, generator=g)))
    lin_all = torch.randn(B, C2 + 24, generator=g).to(DEV)
    lin = lin_all[:, 8:8 + C2]
    gain = torch.tensor(0.6, device=DEV)
    ga = ops.conv_igemm(gout, wd, 9, alpha=0.8)
    gr_ref, glin_ref, gg_ref = ops.mod_silu_drop_bwd(r1, lin, gain, ga, pdrop, 99, 3, 4)
    gr, glin, gg = ops.conv3x3_modbwd(gout, wd, 0.8, r1, lin, gain, pdrop, 99, 3, 4)
    assert torch.equal(gr, gr_ref)
    assert rel(glin, glin_ref) <= 1e-5 and abs(gg.item() - gg_ref.item()) <= 1e-4 * (abs(gg_ref.item()) + 1e-3)


@pytest.mark.parametrize("B,H,W,Cin,Cout,pdrop", [(3, 16, 16, 128, 64, 0.25), (2, 4, 8, 64, 72, 0.1), (1, 5, 7, 64, 72, 0.2), (128, 32, 32, 256, 256, 0.13),
                                                   (128, 16, 16, 256, 256, 0.13), (128, 8, 8, 256, 256, 0.13),
                                                   (512, 8, 8, 64, 256, 0.1), (2, 8, 8, 64, 128, 0.0)])
def test_dropout_marks_in_saved_preactivation(ops, B, H, W, Cin, Cout, pdrop):
    """conv3x3_mod(mark_dropped=True) returns u with NaN exactly where the Philox mask (ops.dropout_mask, the stream the
    oracle's gradient-parity test injects) dropped the element and the plain u everywhere else; a2 is unchanged.  With
    that u, conv3x3_modbwd(u_marked=True) -- which regenerates no random stream -- and the separate mod_silu_drop_bwd kernel
    give the results of the unmarked path: gr bit for bit, glin / ggain up to fp32 summation order."""
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = nhwc(q(torch.randn(B, Cin, H, W, generator=g)))
    wp = pack_fwd(q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)))
    lin_all = torch.randn(B, Cout + 40, generator=g).to(DEV)
    lin = lin_all[:, 8:8 + Cout]
    gain = torch.tensor(0.7, device=DEV)
    u, a2 = ops.conv3x3_mod(x, wp, lin, gain, pdrop, 1234, 5, 6)
    um, a2m = ops.conv3x3_mod(x, wp, lin, gain, pdrop, 1234, 5, 6, mark_dropped=True)
    assert torch.equal(a2, a2m)
    keep = ops.dropout_mask(u.numel(), pdrop, 1234, 5, 6, DEV).view_as(u).bool() if pdrop > 0 else torch.ones_like(u, dtype=torch.bool)
    assert torch.equal(torch.isnan(um), ~keep)
    assert torch.equal(torch.where(keep, um, u), u)
    if pdrop > 0:
        assert abs((~keep).float().mean().item() - pdrop) < 0.02
    gout = nhwc(q(torch.randn(B, Cin, H, W, generator=g)))
    wd = pack_fwd(q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)))
    ga = ops.conv_igemm(gout, wd, 9, alpha=0.8)
    gr_ref, glin_ref, gg_ref = ops.mod_silu_drop_bwd(u, lin, gain, ga, pdrop, 1234, 5, 6)
    # the separate backward kernel (layers whose H*W is not a multiple of 32 take it) accepts the marked tensor
    gr2, glin2, gg2 = ops.mod_silu_drop_bwd(um, lin, gain, ga, pdrop, 1234, 5, 6)
    assert torch.equal(gr2, gr_ref) and torch.isfinite(glin2).all() and rel(glin2, glin_ref) <= 1e-5
    assert abs(gg2.item() - gg_ref.item()) <= 1e-4 * (abs(gg_ref.item()) + 1e-3)
    if (H * W) % 32:
        return
    gr, glin, gg = ops.conv3x3_modbwd(gout, wd, 0.8, um, lin, gain, pdrop, 999, 1, 2, u_marked=True)   # (seed / stream unused)
    assert torch.equal(gr, gr_ref)
    assert torch.isfinite(glin).all() and rel(glin, glin_ref) <= 1e-5
    assert abs(gg.item() - gg_ref.item()) <= 1e-4 * (abs(gg_ref.item()) + 1e-3)


@pytest.mark.parametrize("B,H,W,C1,C2", [(2, 8, 8, 64, 128), (3, 5, 7, 128, 72), (128, 32, 32, 256, 256)])
@pytest.mark.parametrize("with_extra", [False, True])
def test_conv3x3_silubwd_epilogue_is_bit_identical(ops, B, H, W, C1, C2, with_extra):
    g = torch.Generator().manual_seed(B + C1 + C2)
    gr = nhwc(q(torch.randn(B, C1, H, W, generator=g)))
    wd = pack_fwd(q(torch.randn(C2, C1, 3, 3, generator=g) / math.sqrt(C1 * 9)))
    xpre = nhwc(q(torch.randn(B, C2, H, W, generator=g)))
    extra = nhwc(q(torch.randn(B, C2, H, W, generator=g))) if with_extra else None
    ref = ops.silu_bwd(xpre, ops.conv_igemm(gr, wd, 9), extra, 0.7)
    got = ops.conv3x3_silubwd(gr, wd, xpre, extra, 0.7)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,H,W,Cin,Cout,taps,version", [
    (2, 8, 8, 64, 64, 9, 1), (3, 5, 7, 128, 72, 1, 1), (4, 16, 16, 256, 256, 1, 1),        # k_conv_igemm (3x3 and 1x1)
    (6, 32, 32, 64, 128, 1, 2), (128, 32, 32, 256, 512, 1, 2),                             # k_conv_igemm2 (the 32x32 1x1 layers)
    (16, 8, 8, 256, 256, 9, 5), (128, 8, 8, 256, 256, 9, 5),                               # k_conv3x3_s
    (4, 32, 32, 64, 128, 9, 6), (5, 16, 16, 64, 192, 9, 6), (128, 32, 32, 256, 256, 9, 6),  # k_conv3x3_v6
    (128, 16, 16, 256, 256, 9, 6),
])
def test_conv_output_descriptor(ops, B, H, W, Cin, Cout, taps, version):
    """edm_conv_igemm_o on every kernel generation that has the form: the strided output (the left column block of a wider
    NHWC buffer), its mp_silu twin and the split output are the plain kernel's result bit for bit, mp_silu of it exactly
    as the elementwise kernel computes it, and no byte outside the destination columns is touched."""
    g = torch.Generator().manual_seed(B + Cin + Cout + taps)
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).to(DEV)
    r = torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16).to(DEV)
    wp = (torch.randn(taps, Cout, Cin, generator=g) / math.sqrt(Cin * taps)).to(torch.bfloat16).to(DEV)
    old = ops.IGEMM_VERSION
    ops.IGEMM_VERSION = version
    try:
        assert ops._KERNEL_ID[ops._igemm_entry(B * H * W, W, Cout, taps, Cin)] == version
        ref = ops.conv_igemm(x, wp, taps, residual=r, alpha=0.8, beta=0.6)
        Cs = 64
        cat = torch.full((B, H, W, Cout + Cs), 7.0, device=DEV, dtype=torch.bfloat16)
        sil = torch.full((B, H, W, Cout + Cs), 7.0, device=DEV, dtype=torch.bfloat16)
        out = ops.conv_igemm(x, wp, taps, residual=r, alpha=0.8, beta=0.6, out=cat[..., :Cout], silu_out=sil[..., :Cout])
        assert out.data_ptr() == cat.data_ptr()
        assert torch.equal(cat[..., :Cout], ref) and torch.equal(sil[..., :Cout], ops.silu_fwd(ref))
        assert bool((cat[..., Cout:] == 7.0).all()) and bool((sil[..., Cout:] == 7.0).all())
        only = torch.full((B, H, W, Cout + Cs), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.conv_igemm(x, wp, taps, residual=r, alpha=0.8, beta=0.6, out=only[..., :Cout])     # strided, no silu twin
        assert torch.equal(only[..., :Cout], ref) and bool((only[..., Cout:] == 7.0).all())
        for c in sorted({8, Cout // 2, Cout - 8}):
            if not (0 < c < Cout and c % 8 == 0):
                continue
            ya = torch.empty(B, H, W, c, device=DEV, dtype=torch.bfloat16)
            yb = torch.empty(B, H, W, Cout - c, device=DEV, dtype=torch.bfloat16)
            ops.conv_igemm(x, wp, taps, residual=r, alpha=0.8, beta=0.6, split=(c, ya, yb))
            assert torch.equal(ya, ref[..., :c]) and torch.equal(yb, ref[..., c:]), c
    finally:
        ops.IGEMM_VERSION = old


@pytest.mark.parametrize("B,H,W,Ci,Cs", [(2, 8, 8, 64, 128), (3, 5, 7, 192, 192), (128, 32, 32, 256, 256)])
def test_skip_half_kernels_match_the_concat_kernels(ops, B, H, W, Ci, Cs):
    """edm_skip_half_fwd / _bwd (the skip half of the decoder's concatenated operands, the half of d loss / d cat that
    belongs to the gated skip) against the standalone concat kernels they replace: bit for bit."""
    g = torch.Generator().manual_seed(B + Ci + Cs)
    inp = torch.randn(B, H, W, Ci, generator=g).to(torch.bfloat16).to(DEV)
    skip = torch.randn(B, H, W, Cs, generator=g).to(torch.bfloat16).to(DEV)
    gate = torch.rand(B, Cs, generator=g).to(DEV)
    cat_ref, sil_ref = ops.concat_gate_fwd(inp, skip, gate, True)
    cat = torch.empty_like(cat_ref)
    sil = torch.empty_like(cat_ref)
    cat[..., :Ci] = inp
    sil[..., :Ci] = ops.silu_fwd(inp)
    ops.skip_half_fwd(skip, gate, cat, sil)
    assert torch.equal(cat, cat_ref) and torch.equal(sil, sil_ref)
    gcat = torch.randn(B, H, W, Ci + Cs, generator=g).to(torch.bfloat16).to(DEV)
    gmean = torch.randn(B, Cs, generator=g).to(DEV)
    ginp_ref, gskip_ref = ops.concat_gate_bwd(gcat, gate, gmean, Ci)
    gskip = ops.skip_half_bwd(gcat[..., Ci:].contiguous(), gate, gmean)
    assert torch.equal(gskip, gskip_ref) and torch.equal(ginp_ref, gcat[..., :Ci])


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
        extra = q(torch.randn(x.shape, generator=g))          # gradient of the same tensor along the U-Net skip
        gx2 = ops.pixelnorm_silu_bwd(xn, d, nhwc(gxn), 0.8, nhwc(ga), gadd=nhwc(extra))
        close_bf16(nchw(gx2), xr.grad + extra, l2=8e-3, mx=3e-2)


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
    big = q(torch.randn(2, 64, 16, 16, generator=g))
    close_bf16(nchw(ops.up2(nhwc(x), 0.25, add=nhwc(big))),
               0.25 * F.interpolate(x, scale_factor=2, mode="nearest-exact") + big)


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
    # the first half alone (edm_mod_silu_drop_bwd_raw): the raw modulation gradient into a column slice of a wider buffer --
    # what a block does when the network shares one buffer and ONE edm_mod_finish_multi finishes all blocks; an odd map size
    # (H*W = 49, no multiple of 32: the case the fused dgrad epilogue refuses)
    gm_all = torch.zeros(B, C + 96, device=DEV)
    gr2, none1, none2 = ops.mod_silu_drop_bwd(rd, lin.to(DEV), gain.to(DEV), nhwc(ga), pdrop, seed, sub, step,
                                              gm_out=gm_all[:, 32:32 + C])
    assert none1 is None and none2 is None and torch.equal(gr2, gr)
    assert (gm_all[:, :32] == 0).all() and (gm_all[:, 32 + C:] == 0).all()
    gm = gm_all[:, 32:32 + C].cpu()
    assert rel(gm * gain, ll.grad) < 2e-3                                   # glin = gm * gain (k_mod_finish)
    assert abs((gm * lin).sum().item() - gg.grad.item()) <= 2e-3 * abs(gg.grad.item()) + 1e-3
    r7 = nhwc(q(torch.randn(B, C, 7, 7, generator=g)))
    ga7 = nhwc(q(torch.randn(B, C, 7, 7, generator=g)))
    gr_a, glin_a, gg_a = ops.mod_silu_drop_bwd(r7, lin.to(DEV), gain.to(DEV), ga7, pdrop, seed, sub, step)
    gm7 = torch.zeros(B, C, device=DEV)
    gr_b, _, _ = ops.mod_silu_drop_bwd(r7, lin.to(DEV), gain.to(DEV), ga7, pdrop, seed, sub, step, gm_out=gm7)
    assert torch.equal(gr_a, gr_b) and rel((gm7 * gain.to(DEV)).cpu(), glin_a.cpu()) < 1e-5


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


@pytest.mark.parametrize("B,Ci,Cs,H,W", [(2, 64, 128, 8, 8), (3, 192, 192, 5, 7), (128, 256, 256, 32, 32), (4, 64, 768, 16, 16),
                                          (2, 32, 1024, 4, 4),
                                          # batches past one 64-KiB LDS chunk of the weight-gradient pass (B * (R + 4) > 16384):
                                          (1024, 64, 256, 4, 4), (400, 64, 768, 2, 2)])
def test_skip_gate_fused(ops, B, Ci, Cs, H, W):
    """edm_skip_gate_fwd / _bwd (mean over H*W + gate MLP and their backward in ONE launch each, one workgroup per sample)
    against the two-launch path (k_reduce_hw_det + k_scalelong_*) it replaces -- bit for bit in the forward (same fixed
    summation tree is NOT required: compared at 1e-6) -- and against fp64 torch math; deterministic across calls."""
    g = torch.Generator().manual_seed(B + Cs + H)
    R = Cs // 16
    skd = torch.randn(B, H, W, Cs, generator=g).to(torch.bfloat16).to(DEV)
    gcd = torch.randn(B, H, W, Ci + Cs, generator=g).to(torch.bfloat16).to(DEV)
    w1h = (torch.randn(R, Cs + 1, generator=g) / math.sqrt(Cs + 1)).to(DEV)
    w2h = (torch.randn(Cs, R, generator=g) / math.sqrt(R)).to(DEV)
    mean, gate, z1 = ops.skip_gate_fwd(skd, w1h, w2h)
    mean2, gate2, z12 = ops.skip_gate_fwd(skd, w1h, w2h)
    assert torch.equal(mean, mean2) and torch.equal(gate, gate2) and torch.equal(z1, z12)      # bit-reproducible
    mean_ref = skd.double().mean(dim=(1, 2))
    assert rel(mean, mean_ref) < 1e-6
    m1 = torch.cat([mean_ref, torch.ones(B, 1, device=DEV, dtype=torch.float64)], 1)
    z1_ref = m1 @ w1h.double().t()
    gate_ref = torch.sigmoid((torch.nn.functional.silu(z1_ref) / 0.596) @ w2h.double().t())
    assert rel(z1, z1_ref) < 1e-5 and rel(gate, gate_ref) < 1e-5
    mean_o = ops.reduce_hw(skd, scale=1.0 / (H * W))
    gate_o, z1_o = ops.scalelong_fwd(mean_o, w1h, w2h)
    assert rel(mean, mean_o) < 1e-6 and rel(gate, gate_o) < 1e-6 and rel(z1, z1_o) < 1e-6
    # backward
    gmean, gw1, gw2 = ops.skip_gate_bwd(gcd, Ci, skd, mean, w1h, w2h, gate, z1)
    ggate_o = ops.reduce_hw(gcd, C=Cs, c_off=Ci, y=skd)
    gmean_o, gw1_o, gw2_o = ops.scalelong_bwd(mean_o, w1h, w2h, gate_o, z1_o, ggate_o)
    assert rel(gmean, gmean_o) < 1e-5 and rel(gw1, gw1_o) < 1e-4 and rel(gw2, gw2_o) < 1e-4
    ggate_ref = (gcd[..., Ci:].double() * skd.double()).sum(dim=(1, 2))
    gz2 = ggate_ref * gate_ref * (1 - gate_ref)
    gw2_ref = gz2.t() @ (torch.nn.functional.silu(z1_ref) / 0.596)
    assert rel(gw2, gw2_ref) < 1e-4


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


def _qkv_perm(C, heads):
    """packed channel (head, which, dd) -> reference channel head*3d + dd*3 + which (networks.py:194)."""
    d = C // heads
    idx = torch.empty(3 * C, dtype=torch.long)
    for h in range(heads):
        for w in range(3):
            for dd in range(d):
                idx[h * 3 * d + w * d + dd] = h * 3 * d + dd * 3 + w
    return idx


@pytest.mark.parametrize("B,H,W,heads,hd", [
    (2, 16, 16, 4, 64), (3, 8, 8, 2, 64), (2, 4, 4, 1, 64), (1, 14, 14, 2, 64), (2, 7, 7, 1, 64),
    # head dims of the other reference configs (streamed-operand kernels, attention_generic.hip):
    # MNIST 512/4 @7x7, ImageNet-64 576/4 @16x16 and 768/4 @8x8, plus ragged token counts
    (2, 7, 7, 2, 128), (1, 16, 16, 2, 144), (2, 8, 8, 2, 192), (1, 14, 14, 1, 144), (2, 5, 5, 1, 32), (1, 12, 12, 1, 192),
    (1, 16, 16, 1, 128)])
def test_attention_fwd_bwd(ops, B, H, W, heads, hd):
    g = torch.Generator().manual_seed(B + H + heads + hd)
    C = hd * heads
    N = H * W
    qkv_ref_layout = q(torch.randn(B, 3 * C, H, W, generator=g))          # reference channel order
    gy = q(torch.randn(B, C, H, W, generator=g))
    x = qkv_ref_layout.clone().requires_grad_(True)
    t = x.view(B, heads, hd, 3, N)
    t = O.q_bf16(O.rms_div(t, [2]))
    qq, kk, vv = t.unbind(3)
    s = torch.einsum("bhdi,bhdj->bhij", qq, kk) / math.sqrt(hd)
    p = O.q_bf16(torch.softmax(s, dim=-1))
    y_ref = torch.einsum("bhij,bhdj->bhdi", p, vv).reshape(B, C, H, W)
    y_ref.backward(gy)
    perm = _qkv_perm(C, heads)
    qkv_packed = nhwc(qkv_ref_layout[:, perm])
    y = ops.attention_fwd(qkv_packed, heads)
    close_bf16(nchw(y), y_ref.detach(), l2=6e-3, mx=3e-2)
    gqkv = ops.attention_bwd(qkv_packed, y, nhwc(gy), heads)
    close_bf16(nchw(gqkv), x.grad[:, perm], l2=1.5e-2, mx=6e-2)


def test_linear_and_embedding(ops):
    g = torch.Generator().manual_seed(12)
    B, Fd, E, K = 5, 64, 256, 10
    X, Wt = torch.randn(B, Fd, generator=g), torch.randn(E, Fd, generator=g)
    dY = torch.randn(B, E, generator=g)
    assert rel(ops.linear_fwd(X.to(DEV), Wt.to(DEV)).cpu(), X @ Wt.t()) < 1e-5
    assert rel(ops.linear_dgrad(dY.to(DEV), Wt.to(DEV)).cpu(), dY @ Wt) < 1e-5
    assert rel(ops.linear_wgrad(dY.to(DEV), X.to(DEV)).cpu(), dY.t() @ X) < 1e-5
    ecfg = O.EmbeddingCfg(Fd, E, K)
    P = {"embedding.fourier_embed.freqs": 2 * math.pi * torch.randn(Fd, generator=g),
         "embedding.fourier_embed.phases": 2 * math.pi * torch.rand(Fd, generator=g),
         "embedding.sigma_embed.weight": torch.randn(E, Fd, generator=g),
         "embedding.class_embed.linear.weight": torch.randn(E, K, generator=g)}
    sigma = torch.randn(B, generator=g).exp()
    labels = torch.randint(0, K, (B,), generator=g)
    for k_ in ("embedding.sigma_embed.weight", "embedding.class_embed.linear.weight"):
        P[k_].requires_grad_(True)
    four_ref, out_ref = O.embedding_forward(P, ecfg, sigma, labels)
    gout = torch.randn(B, E, generator=g)
    out_ref.backward(gout)
    four = ops.fourier_fwd(sigma.to(DEV), P["embedding.fourier_embed.freqs"].to(DEV),
                           P["embedding.fourier_embed.phases"].to(DEV), B)
    assert (four.cpu() - four_ref).abs().max() < 2e-4                    # fp32 cos of a large argument
    wsh = O.effective_weight(P["embedding.sigma_embed.weight"].detach()).to(DEV)
    wch = O.effective_weight(P["embedding.class_embed.linear.weight"].detach()).to(DEV)
    es = ops.linear_fwd(four, wsh)
    pre, out = ops.embed_combine_fwd(es, wch, labels.to(DEV), 0.5)
    assert (out.cpu() - out_ref.detach()).abs().max() < 2e-4
    ges, gwch = ops.embed_combine_bwd(gout.to(DEV), pre, labels.to(DEV), 0.5, wch.shape)
    gwsh = ops.linear_wgrad(ges, four)
    gws = ops.wgrad_finish(gwsh.view(1, 1, E, Fd), P["embedding.sigma_embed.weight"].detach().to(DEV), 1, Fd)
    gwc = ops.wgrad_finish(gwch.view(1, 1, E, K), P["embedding.class_embed.linear.weight"].detach().to(DEV), 1, K)
    assert rel(gws.cpu(), P["embedding.sigma_embed.weight"].grad) < 2e-3
    assert rel(gwc.cpu(), P["embedding.class_embed.linear.weight"].grad) < 2e-3
    # unconditional + scalar sigma
    _, out_u = O.embedding_forward(P, ecfg, torch.tensor(1.7), None)
    four_u = ops.fourier_fwd(torch.tensor([1.7], device=DEV), P["embedding.fourier_embed.freqs"].to(DEV),
                             P["embedding.fourier_embed.phases"].to(DEV), 1)
    _, o_u = ops.embed_combine_fwd(ops.linear_fwd(four_u, wsh), None, None, 0.5)
    assert (o_u.cpu() - out_u.detach()).abs().max() < 2e-4


def test_step_level_kernels(ops):
    g = torch.Generator().manual_seed(13)
    B = 6
    clean = 0.5 * torch.randn(B, 3, 8, 8, generator=g)
    eps, noise = torch.randn(B, generator=g), torch.randn(B, 3, 8, 8, generator=g)
    noisy_ref, sigma_ref = O.diffuse(clean, eps, noise, -1.2, 1.2)
    noisy, sigma = ops.diffuse_given(clean.to(DEV), eps.to(DEV), noise.to(DEV), -1.2, 1.2)
    assert torch.allclose(noisy.cpu(), noisy_ref, rtol=1e-5, atol=1e-5) and torch.allclose(sigma.cpu(), sigma_ref, rtol=1e-5)
    # Philox diffuser: distribution of ln(sigma) and of the unit noise (RNG parity is distributional)
    big = torch.zeros(4096, 3, 8, 8, device=DEV)
    nz, sg = ops.diffuse(big, -1.2, 1.2, 42, 3)
    ls = sg.log()
    assert abs(ls.mean().item() + 1.2) < 0.08 and abs(ls.std().item() - 1.2) < 0.08
    unit = nz / sg.view(-1, 1, 1, 1)
    assert abs(unit.mean().item()) < 0.01 and abs(unit.std().item() - 1) < 0.01
    nz2, sg2 = ops.diffuse(big, -1.2, 1.2, 42, 3)
    assert torch.equal(nz, nz2) and torch.equal(sg, sg2)                 # counter-based: replayable
    nz3, _ = ops.diffuse(big, -1.2, 1.2, 42, 4)
    assert not torch.equal(nz, nz3)
    # loss (closed form of the reference's own test) + gradient
    D = torch.randn(B, 3, 8, 8, generator=g).requires_grad_(True)
    w = O.loss_weight(sigma_ref, 0.5)
    ref = torch.mean(w.view(-1, 1, 1, 1) * (D - clean) ** 2)
    ref.backward()
    loss, dD = ops.weighted_mse(D.detach().to(DEV), clean.to(DEV), sigma_ref.to(DEV), 0.5)
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert rel(dD.cpu(), D.grad) < 1e-5
    # Adam + EMA against the oracle
    n = 1003
    th = torch.randn(n, generator=g)
    m, v, ema = torch.zeros(n), torch.zeros(n), th.clone()
    thd, md, vd, ed = th.clone().to(DEV), m.clone().to(DEV), v.clone().to(DEV), ema.clone().to(DEV)
    gam = O.sigma_rel_to_gamma(0.13)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g)
        O.adam_step(th, gr, m, v, step, 0.02)
        beta = O.ema_beta(step - 1, gam)
        O.ema_step(ema, th, beta)
        ops.adam_ema(thd, gr.to(DEV), md, vd, ed, 0.02, 0.9, 0.999, 1e-8, step, beta)
        assert torch.allclose(thd.cpu(), th, rtol=1e-5, atol=1e-6)
        assert torch.allclose(ed.cpu(), ema, rtol=1e-5, atol=1e-6)
    # Heun updates
    x, Dn, D1 = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    t0, t1 = 3.0, 1.7
    dx_ref = (x - Dn) / t0
    x1_ref = x + (t1 - t0) * dx_ref
    dx, x1 = ops.heun_euler(x.to(DEV), Dn.to(DEV), t0, t1)
    assert torch.allclose(dx.cpu(), dx_ref, atol=1e-6) and torch.allclose(x1.cpu(), x1_ref, atol=1e-6)
    out = ops.heun_correct(x.to(DEV), dx, x1, D1.to(DEV), t0, t1)
    assert torch.allclose(out.cpu(), x + (t1 - t0) * (0.5 * dx_ref + 0.5 * (x1_ref - D1) / t1), atol=1e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 64, 64, 64, 64), (2, 33, 63, 64, 128), (1, 16, 100, 64, 64),
                                             (1, 9, 126, 96, 64), (3, 5, 70, 64, 64)])
def test_conv_wgrad_wide_images(ops, B, H, W, Cin, Cout):
    """3x3 weight gradient on images wider than one 64-row stage (W 63..126: the 8-slot rolling window of
    conv_wgrad2.hip, used by the 64x64 layers of the ImageNet / latent nets) vs an fp64 convolution backward."""
    g = torch.Generator().manual_seed(W + Cin)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    gy = q(torch.randn(B, Cout, H, W, generator=g))
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, padding=1).backward(gy.double())
    ref = w.grad.permute(2, 3, 0, 1).reshape(9, Cout, Cin)
    assert ops.WGRAD_VERSION == 2
    slabs = ops.conv_wgrad(nhwc(x), nhwc(gy), 9)
    got = slabs.double().sum(0).cpu()
    assert rel(got, ref) <= 1e-5, f"wide wgrad rel {rel(got, ref):.3e}"


@pytest.mark.parametrize("B,H,W,C", [(3, 8, 8, 64), (2, 6, 10, 256), (5, 4, 4, 768), (1, 16, 16, 40)])
def test_resample_fused_into_neighbours_is_bit_identical(ops, B, H, W, C):
    """Round 6: the 2x2 average pool of an EncD block inside the pixel-norm kernels (forward AND backward) and a DecU block's
    upsample that also emits mp_silu of its result -- against the separate kernels they replace, bit for bit."""
    g = torch.Generator().manual_seed(B + H + C)
    x = torch.randn(B, 2 * H, 2 * W, C, generator=g).to(torch.bfloat16).to(DEV)
    xn, a, d = ops.pool_pixelnorm_silu_fwd(x)
    xn0, a0, d0 = ops.pixelnorm_silu_fwd(ops.pool2(x, 0.25))
    assert torch.equal(xn, xn0) and torch.equal(a, a0) and torch.equal(d, d0)
    gxn = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    ga = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    gadd = torch.randn(B, 2 * H, 2 * W, C, generator=g).to(torch.bfloat16).to(DEV)
    for add in (None, gadd):
        for gx_in in (gxn, None):
            gx = ops.pool_pixelnorm_silu_bwd(xn, d, gx_in, 0.7, ga, gadd=add)
            gx0 = ops.up2(ops.pixelnorm_silu_bwd(xn0, d0, gx_in, 0.7, ga), 0.25, add=add)
            assert torch.equal(gx, gx0)
    xs = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    y, s = ops.up2_silu(xs)
    y0 = ops.up2(xs)
    assert torch.equal(y, y0) and torch.equal(s, ops.silu_fwd(y0))


@pytest.mark.parametrize("B,H,W,Cin,Cout,C2,imgs", [
    (128, 32, 32, 256, 256, 512, [0, 63, 64, 127]),      # 512 x 128 tiles, images aligned to the tiles (static borders)
    (128, 16, 16, 256, 256, 512, [0, 1, 126, 127]),      # 512 x 64 tiles, two images per tile
    (99, 24, 24, 128, 192, 320, [0, 49, 98]),            # per-lane border bits, ragged last tile, Cout = 3 tiles of 64
    (512, 8, 8, 256, 256, 768, [0, 255, 256, 511])])     # the sampler's 8x8 layers
@pytest.mark.parametrize("dest", [False, True])
def test_conv3x3_fold_skip_projection(ops, B, H, W, Cin, Cout, C2, imgs, dest):
    """Round 6: Y = alpha3 * conv3x3(x, W3) + alpha1 * conv1x1(x2, W1) in one launch of the static-schedule kernel (the decoder
    block's skip projection behind the nine taps of its second 3x3 conv, networks.py:313, 325-327) against fp64 on the same
    bf16-rounded operands; with the strided output + mp_silu form of the copy-free concat; and against the unfused pair, which
    differs only by the bf16 rounding of the projection's result."""
    assert ops.conv3x3_fold_supported((B, H, W, Cin), Cout, C2)
    g = torch.Generator().manual_seed(B + H + Cin + C2)
    x = torch.randn(B, Cin, H, W, generator=g)
    x2 = torch.randn(B, C2, H, W, generator=g)
    w3 = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    w1 = torch.randn(Cout, C2, 1, 1, generator=g) / math.sqrt(C2)
    a3, a1 = 0.39, 0.92
    xh, x2h = nhwc(x), nhwc(x2)
    if dest:
        Ct = Cout + 64
        cat = torch.zeros(B, H, W, Ct, device=DEV, dtype=torch.bfloat16)
        sil = torch.zeros_like(cat)
        from tinyedm_amd.networks import _col_block
        y = ops.conv3x3_fold(xh, pack_fwd(w3), x2h, pack_fwd(w1), a3, a1, out=_col_block(cat, Cout), silu_out=_col_block(sil, Cout))
        assert y.data_ptr() == cat.data_ptr() and float(cat[..., Cout:].abs().max()) == 0.0 and float(sil[..., Cout:].abs().max()) == 0.0
        ysil = sil[..., :Cout].contiguous()
        assert torch.equal(ysil, ops.silu_fwd(cat[..., :Cout].contiguous()))
        y = cat[..., :Cout].contiguous()
    else:
        y = ops.conv3x3_fold(xh, pack_fwd(w3), x2h, pack_fwd(w1), a3, a1)
    ref = a3 * F.conv2d(q(x[imgs]).double(), q(w3).double(), padding=1) + a1 * F.conv2d(q(x2[imgs]).double(), q(w1).double())
    close_bf16(nchw(y)[imgs], ref)
    if not dest:
        y0 = ops.conv_igemm(xh, pack_fwd(w3), 9, residual=ops.conv_igemm(x2h, pack_fwd(w1), 1), alpha=a3, beta=a1)
        assert rel(y.float(), y0.float()) <= 4e-3


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(128, 32, 32, 256, 256),     # training batch: two tiles per workgroup
                                             (130, 32, 32, 64, 256),      # 260 pixel tiles: two or three per workgroup
                                             (512, 16, 16, 256, 256),     # the samplers' 16x16 layers (two images per tile)
                                             (200, 32, 32, 64, 384)])     # three channel tiles: 240 workgroups, ragged walk
def test_persistent_conv3x3_walks_its_tiles_like_single_tile_workgroups(ops, B, H, W, Cin, Cout):
    """k_conv3x3_v6's persistent form (round 6, opt-in: a workgroup walks several pixel tiles, the next tile's slab and first
    weight tiles landing under the epilogue) against the same kernel with one tile per workgroup (EDM_V6_PERSIST, read per
    call): bit-identical outputs for the epilogues it is built for -- plain (+ residual, + the strided / silu output
    descriptor), forward modulation, the split-bf16 fp32 form -- and, on sampled images, against the fp64 convolution."""
    import os
    from tinyedm_amd import _lib
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = nhwc(q(torch.randn(B, Cin, H, W, generator=g)))
    wf = q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    wp = pack_fwd(wf)
    res = nhwc(q(torch.randn(B, Cout, H, W, generator=g)))
    lin = torch.randn(B, Cout + 24, generator=g).to(DEV)[:, 8:8 + Cout]
    gain = torch.tensor(0.6, device=DEV)
    wide = torch.zeros(B, H, W, Cout + 64, device=DEV, dtype=torch.bfloat16)
    wide2 = torch.zeros_like(wide)
    xp = ops.f32_to_pairs(x.float())
    pk = ops.split_pack(wf.reshape(Cout, -1).contiguous().to(DEV), 9) if Cin <= 256 else None

    def run():
        out = {}
        out["plain"] = ops.conv_igemm(x, wp, 9, alpha=0.9)
        out["res"] = ops.conv_igemm(x, wp, 9, residual=res, alpha=0.7, beta=0.6)
        wide.zero_(); wide2.zero_()
        ops.conv_igemm(x, wp, 9, out=wide[..., :Cout], silu_out=wide2[..., :Cout])
        out["desc"], out["desc_silu"] = wide.clone(), wide2.clone()
        out["u"], out["a2"] = ops.conv3x3_mod(x, wp, lin, gain, 0.13, 77, 1, 2)
        if pk is not None:
            out["split"] = ops.split_conv(xp, pk, 9, alpha=0.8)
        return out

    prev = os.environ.get("EDM_V6_PERSIST")
    try:
        os.environ["EDM_V6_PERSIST"] = "0"
        ref = run()
        os.environ["EDM_V6_PERSIST"] = "1"
        n0 = _lib.call("edm_v6_persistent_launches")
        got = run()
        assert _lib.call("edm_v6_persistent_launches") - n0 >= 5       # the form under test is the one that ran
    finally:
        if prev is None:
            os.environ.pop("EDM_V6_PERSIST", None)
        else:
            os.environ["EDM_V6_PERSIST"] = prev
    for k, v in ref.items():
        assert torch.equal(got[k], v), k
    for b in (0, B // 2, B - 1):       # and against the definition
        want = 0.9 * F.conv2d(x[b:b + 1].permute(0, 3, 1, 2).double(), wf.double().to(DEV), padding=1)
        assert rel(got["plain"][b:b + 1].permute(0, 3, 1, 2).double(), want) <= 6e-3


def test_skip_gate_fwd_multi_matches_the_per_tensor_launches(ops):
    """edm_skip_gate_fwd_multi (round 6: the ScaleLong gates of every decoder block from ONE launch): mean, gate and z1 of each
    tensor bit for bit those of edm_skip_gate_fwd (same body per (tensor, sample) workgroup), for tensors of different map
    sizes, batch sizes and hidden widths sharing a channel count; a mixed channel count is refused."""
    g = torch.Generator().manual_seed(77)
    C = 256
    items = []
    for B, H, W, R in [(128, 32, 32, 16), (128, 16, 16, 16), (5, 8, 8, 24), (3, 5, 7, 8), (1, 1, 1, 16)]:
        skip = nhwc(q(torch.randn(B, C, H, W, generator=g)))
        w1h = (torch.randn(R, C + 1, generator=g) / math.sqrt(C + 1)).to(DEV)
        w2h = (torch.randn(C, R, generator=g) / math.sqrt(R)).to(DEV)
        items.append((skip, w1h, w2h))
    got = ops.skip_gate_fwd_multi(items)
    for (skip, w1h, w2h), (mean, gate, z1) in zip(items, got):
        m0, g0, z0 = ops.skip_gate_fwd(skip, w1h, w2h)
        assert torch.equal(mean, m0) and torch.equal(gate, g0) and torch.equal(z1, z0)
        ref_mean = skip.float().mean(dim=(1, 2))
        assert rel(mean, ref_mean) <= 1e-5
    # ... and with the (cat, sil) buffers of the consuming decoder blocks: the right halves == skip_half_fwd, the left halves
    # (the producer's columns) untouched; an item without buffers rides in the same launch
    Ci = 64
    bufs = []
    for k, (skip, _, _) in enumerate(items):
        B, H, W, _ = skip.shape
        cat = torch.full((B, H, W, Ci + C), 3.0, device=DEV, dtype=torch.bfloat16)
        bufs.append(None if k == 2 else (cat, None if k == 1 else torch.full_like(cat, 5.0)))
    got2 = ops.skip_gate_fwd_multi([it + (b if b is not None else ()) for it, b in zip(items, bufs)])
    for (skip, _, _), (mean, gate, z1), (m0, g0, z0), b in zip(items, got2, got, bufs):
        assert torch.equal(mean, m0) and torch.equal(gate, g0) and torch.equal(z1, z0)
        if b is None:
            continue
        B, H, W, _ = skip.shape
        cat0 = torch.full((B, H, W, Ci + C), 3.0, device=DEV, dtype=torch.bfloat16)
        sil0 = torch.full_like(cat0, 5.0) if b[1] is not None else None
        ops.skip_half_fwd(skip, gate, cat0, sil0)
        assert torch.equal(b[0], cat0)
        if sil0 is not None:
            assert torch.equal(b[1], sil0)
    other = nhwc(q(torch.randn(2, 128, 4, 4, generator=g)))
    with pytest.raises(ValueError):
        ops.skip_gate_fwd_multi([items[0], (other, torch.zeros(8, 129, device=DEV), torch.zeros(128, 8, device=DEV))])


def test_skip_gate_backward_multi_matches_the_per_tensor_launches(ops):
    """edm_skip_gate_bwd_multi + edm_skip_half_bwd_multi (round 6: the deferred backward of every decoder gate, two launches):
    gmean, the per-sample scratch ws and the skip gradient of each gate bit for bit those of edm_skip_gate_bwd (deferred
    weight-gradient form) + edm_skip_half_bwd, for different map / batch sizes and hidden widths."""
    g = torch.Generator().manual_seed(78)
    C = 256
    gates, ref = [], []
    for B, H, W, R in [(128, 32, 32, 16), (64, 16, 16, 16), (5, 8, 8, 24), (3, 5, 7, 8)]:
        skip = nhwc(q(torch.randn(B, C, H, W, generator=g)))
        gcs = nhwc(q(torch.randn(B, C, H, W, generator=g)))
        w1h = (torch.randn(R, C + 1, generator=g) / math.sqrt(C + 1)).to(DEV)
        w2h = (torch.randn(C, R, generator=g) / math.sqrt(R)).to(DEV)
        mean, gate, z1 = ops.skip_gate_fwd(skip, w1h, w2h)
        gmean, ws = ops.skip_gate_bwd(gcs, 0, skip, mean, w1h, w2h, gate, z1, defer_wgrad=True)
        ref.append((gmean, ws, ops.skip_half_bwd(gcs, gate, gmean)))
        gates.append((gcs, 0, skip, w1h, w2h, gate, z1))
    got = ops.skip_gate_bwd_multi(gates)
    holders = [torch.full_like(gt[0], float("nan")) for gt in gates]
    ops.skip_half_bwd_multi([(gt[0], gt[5], gm, h) for gt, (gm, _), h in zip(gates, got, holders)])
    for (gmean, ws), h, (gmean0, ws0, gskip0) in zip(got, holders, ref):
        assert torch.equal(gmean, gmean0) and torch.equal(ws, ws0)
        assert torch.equal(h, gskip0)
    # ... and the one-launch form: the gate's workgroup writes the skip gradient of its sample itself
    holders2 = [torch.full_like(gt[0], float("nan")) for gt in gates]
    got2 = ops.skip_gate_bwd_multi([gt + (h,) for gt, h in zip(gates, holders2)])
    for (gmean, ws), h, (gmean0, ws0, gskip0) in zip(got2, holders2, ref):
        assert torch.equal(gmean, gmean0) and torch.equal(ws, ws0)
        assert torch.equal(h, gskip0)
