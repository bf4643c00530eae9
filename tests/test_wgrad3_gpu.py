"""Grouped stream-K 3x3 weight gradient (csrc/conv_wgrad3.hip, edm_wgrad3_group) against the oracle: the fp64
autograd gradient of F.conv2d(x, effective_weight(w)) w.r.t. the master weight (reference networks.py:32-37), on
the same bf16-rounded operands.  Groups mix layer shapes so that workgroup ranges cross tile and layer boundaries."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import edm_oracle as O
from parity_log import record

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tinyedm_amd import ops as _ops
    return _ops


def q(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _layer(g, B, H, W, Cin, Cout, I=None, perm=False, scale=1.0, accumulate=False):
    """one layer: inputs on the device + the oracle gradient of the master weight"""
    I = Cin if I is None else I
    x = q(torch.randn(B, Cin, H, W, generator=g))
    if I < Cin:
        x[:, I:] = 0.0                      # zero-padded input channels (conv_in: 4 real channels of 32)
    gy = q(torch.randn(B, Cout, H, W, generator=g))
    wm = O.weight_normalize(torch.randn(Cout, I, 3, 3, generator=g) * 1.3)
    p = torch.randperm(Cout, generator=g) if perm else None
    # oracle: packed output channel r is master output channel p[r]
    gy_master = gy
    if p is not None:
        gy_master = torch.empty_like(gy)
        gy_master[:, p] = gy
    w64 = wm.double().clone().requires_grad_(True)
    y = F.conv2d(x[:, :I].double(), O.effective_weight(w64), padding=1)
    y.backward(gy_master.double())
    ref = w64.grad * scale
    g0 = torch.randn(Cout, I, 3, 3, generator=g) if accumulate else torch.zeros(Cout, I, 3, 3)
    grad = g0.clone().to(DEV)
    item = (nhwc(x), nhwc(gy), wm.to(DEV), grad, None if p is None else p.to(torch.int32).to(DEV), scale, accumulate)
    return item, ref + g0.double()


GROUP_SMALL = [
    dict(B=2, H=8, W=8, Cin=64, Cout=64),
    dict(B=3, H=16, W=16, Cin=128, Cout=192, scale=0.7),
    dict(B=1, H=32, W=32, Cin=64, Cout=128, accumulate=True),
    dict(B=5, H=7, W=7, Cin=64, Cout=64),
    dict(B=2, H=8, W=8, Cin=512, Cout=256, perm=True),
    dict(B=2, H=14, W=14, Cin=32, Cout=64, I=4),
    dict(B=4, H=16, W=16, Cin=256, Cout=256, scale=0.5, accumulate=True),
    dict(B=1, H=3, W=3, Cin=32, Cout=32),
]
GROUP_WIDE = [
    dict(B=1, H=64, W=64, Cin=64, Cout=64),
    dict(B=2, H=33, W=63, Cin=64, Cout=128, accumulate=True),
    dict(B=1, H=16, W=100, Cin=64, Cout=64),
    dict(B=1, H=9, W=126, Cin=96, Cout=64, perm=True),
    dict(B=3, H=5, W=70, Cin=64, Cout=160),
]


# channel counts of the default (ImageNet) Denoiser: 3 x 6, 5 x 9, 5 x 12 and 3 x 3 tiles -- the team shapes 1 x 8 and 2 x 4 the
# round-6 plan picks where they leave fewer members idle than the power-of-two rule (conv_wgrad3.hip make_plan)
GROUP_TEAMS = [
    dict(B=2, H=8, W=8, Cin=384, Cout=384),
    dict(B=1, H=8, W=8, Cin=576, Cout=576, scale=0.7),
    dict(B=2, H=4, W=4, Cin=768, Cout=576, perm=True),
    dict(B=1, H=8, W=8, Cin=192, Cout=384, accumulate=True),
]


# layers whose tiles do not fill a team of eight: the plan splits their K range into shares (virtual tiles, conv_wgrad3.hip):
# 1 x 2 tiles (MNIST's 128 channels), 2 x 3 (192), 1 x 1, 3 x 6 with a perm, a share whose last stages lie past the end of K
# (stage counts that are no multiple of the share count), beside a 256-channel layer that must not split
GROUP_KSPLIT = [
    dict(B=9, H=28, W=28, Cin=128, Cout=128, scale=0.7),
    dict(B=16, H=32, W=32, Cin=192, Cout=192, accumulate=True),
    dict(B=40, H=14, W=14, Cin=64, Cout=64),
    dict(B=4, H=16, W=16, Cin=256, Cout=256),
    dict(B=48, H=16, W=16, Cin=384, Cout=384, perm=True),
    dict(B=11, H=28, W=28, Cin=32, Cout=128, I=1),
]
GROUP_KSPLIT_WIDE = [
    dict(B=3, H=64, W=64, Cin=192, Cout=192),
    dict(B=2, H=64, W=64, Cin=64, Cout=128, perm=True, accumulate=True),
]


def test_wgrad3_ksplit_groups_really_split(ops):
    for group in (GROUP_KSPLIT, GROUP_KSPLIT_WIDE):
        ks = ops.wgrad3_plan_ksplit([(kw["B"], kw["H"], kw["W"], kw["Cin"], kw["Cout"]) for kw in group])
        assert sum(k > 1 for k in ks) >= len(group) - 1, ks
    assert ops.wgrad3_plan_ksplit([(4, 16, 16, 256, 256)]) == [1]


@pytest.mark.parametrize("name,group", [("ksplit", GROUP_KSPLIT), ("ksplit-wide", GROUP_KSPLIT_WIDE), ("small", GROUP_SMALL), ("wide", GROUP_WIDE), ("single", GROUP_SMALL[1:2]),
                                        ("sixteen", (GROUP_SMALL * 2)[:16]), ("forty-three", (GROUP_SMALL * 6)[:43]),
                                        ("teams", GROUP_TEAMS)])
def test_wgrad3_group_matches_oracle(ops, name, group):
    g = torch.Generator().manual_seed(len(group) * 31 + 5)
    items, refs = zip(*[_layer(g, **kw) for kw in group])
    ops.wgrad3_group(list(items))
    torch.cuda.synchronize()
    worst = 0.0
    for k, (it, ref) in enumerate(zip(items, refs)):
        got = it[3].cpu()
        assert torch.isfinite(got).all()
        e = rel(got, ref)
        worst = max(worst, e)
        assert e <= 2e-3, f"layer {k} of group {name}: rel {e:.3e}"
    record(f"wgrad3_group[{name}]", worst, 2e-3)


def test_wgrad3_full_size_layers(ops):
    """The shapes the bench dispatches (B=128: 32x32 256->256 and 512->256, 16x16, 8x8) in one group, checked on a
    sub-range of output rows against an fp64 GEMM of the same bf16 operands (unprojected: scale of w_hat folded out
    by giving the kernel a weight whose rows are already orthogonal to the gradient is not possible, so the oracle
    projection is applied to the fp64 raw gradient instead)."""
    g = torch.Generator().manual_seed(99)
    shapes = [(128, 32, 32, 256, 256), (128, 32, 32, 512, 256), (128, 16, 16, 256, 256), (128, 8, 8, 256, 256)]
    items, raws, ws = [], [], []
    for B, H, W, Cin, Cout in shapes:
        x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16)
        gy = torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16)
        wm = O.weight_normalize(torch.randn(Cout, Cin, 3, 3, generator=g))
        grad = torch.zeros(Cout, Cin, 3, 3, device=DEV)
        items.append((x.to(DEV), gy.to(DEV), wm.to(DEV), grad, None, 1.0, False))
        # raw fp64 weight gradient of 16 output rows, on the GPU in fp64 (same bf16 operands)
        rows = torch.arange(0, Cout, Cout // 16)
        xd = x.to(DEV).double().permute(0, 3, 1, 2)
        gd = gy.to(DEV).double().permute(0, 3, 1, 2)[:, rows]
        xu = F.unfold(xd, 3, padding=1)                                  # (B, Cin*9, HW)
        raw = torch.einsum("bon,bkn->ok", gd.flatten(2), xu).view(len(rows), Cin, 3, 3).cpu()
        raws.append((rows, raw))
        ws.append(wm)
        del xd, gd, xu
    ops.wgrad3_group(items)
    torch.cuda.synchronize()
    worst = 0.0
    for (rows, raw), wm, it in zip(raws, ws, items):
        w64 = wm[rows].double().clone().requires_grad_(True)
        (O.effective_weight(w64) * raw).sum().backward()                 # projection through the normalisation
        e = rel(it[3].cpu()[rows], w64.grad)
        worst = max(worst, e)
        assert e <= 2e-3, f"full-size wgrad rel {e:.3e}"
    record("wgrad3_group[full-size B=128]", worst, 2e-3)


def test_wgrad1x1_grouped_launch_matches_per_layer_kernels():
    """edm_conv_wgrad_1x1_group: 1x1 layers of different shapes (ragged tiles, pixel counts that are not a multiple of the
    stage, 16 layers = the maximum) in ONE launch against an fp64 GEMM of the same bf16 operands per layer"""
    from tinyedm_amd import ops
    g = torch.Generator().manual_seed(3)
    shapes = [(3, 5, 7, 160, 96), (2, 16, 16, 256, 768), (5, 8, 8, 512, 256), (1, 3, 3, 32, 32), (16, 16, 16, 128, 320),
              (128, 16, 16, 256, 256), (128, 8, 8, 768, 256), (8, 32, 32, 512, 256)]
    shapes = shapes + shapes                                   # 16 layers
    pairs = []
    for (B, H, W, Cin, Cout) in shapes:
        x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).to("cuda")
        dy = torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16).to("cuda")
        pairs.append((x, dy))
    slabs = ops.conv_wgrad_1x1_group(pairs)
    for (x, dy), sl in zip(pairs, slabs):
        ref = dy.double().reshape(-1, dy.shape[-1]).t() @ x.double().reshape(-1, x.shape[-1])
        got = sl.double().sum(0)[0]
        e = ((got - ref).norm() / ref.norm()).item()
        assert e <= 1e-5, (tuple(x.shape), tuple(dy.shape), e)


def test_wgrad_finish_multi_matches_the_per_tensor_finish():
    """edm_wgrad_finish_multi (one launch for up to 40 small weight gradients; round 6: master row and old gradient loaded
    while the slab loads are in flight) against edm_wgrad_finish per tensor: 1x1 convs / Linears (the prefetched form),
    rows longer than the prefetch window, 3x3 taps, permuted rows, accumulate on and off, ragged Ipad"""
    from tinyedm_amd import ops
    g = torch.Generator().manual_seed(11)
    shapes = [  # (S, taps, O, I, Ipad, perm, accumulate, scale)
        (8, 1, 256, 256, 256, False, False, 1.0), (16, 1, 768, 256, 256, True, True, 0.7), (4, 1, 256, 768, 768, False, True, 1.0),
        (2, 1, 64, 1280, 1280, False, False, 0.5), (3, 9, 32, 4, 32, False, True, 1.0), (1, 1, 10, 3, 8, False, False, 1.0),
        (5, 9, 64, 64, 64, True, False, 1.3), (8, 1, 256, 512, 512, False, True, 1.0),
    ]
    items, refs = [], []
    for (S, taps, O, I, Ipad, perm, acc, scale) in shapes:
        slabs = torch.randn(S, taps, O, Ipad, generator=g).to(DEV)
        w = torch.randn(O, I, taps, generator=g).to(DEV)
        p = torch.randperm(O, generator=g).to(torch.int32).to(DEV) if perm else None
        g0 = torch.randn(O, I, taps, generator=g).to(DEV)
        ref = ops.wgrad_finish(slabs, w, taps, I, perm=p, scale=scale, out=g0.clone() if acc else None)
        grad = g0.clone()
        items.append((slabs, w, grad, p, taps, I, scale, acc))
        refs.append(ref)
    ops.wgrad_finish_multi(items)
    torch.cuda.synchronize()
    for k, (it, ref) in enumerate(zip(items, refs)):
        # (the two kernels may split the S slabs into a different number of partial sums: equal up to fp32 summation order)
        e = rel(it[2], ref)
        assert torch.isfinite(it[2]).all() and e <= 1e-6, f"tensor {k} {shapes[k]}: rel {e:.3e}"
