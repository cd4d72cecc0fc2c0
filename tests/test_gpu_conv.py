"""GPU parity of the implicit-GEMM fp32-MFMA convolution (forward, backward-data, backward-weight)
against torch's CPU conv2d in float64 on the same seeded inputs, for every layer geometry the
DeepLabV2/ResNet-101 model uses (1x1, strided 1x1, dilated 3x3, biased ASPP branches, the 7x7 stem,
the 19-class head) including ragged sizes that exercise tile edges."""
import os
import zlib

import pytest
import torch

from diga_amd import config
import torch.nn.functional as F

from conftest import WINO_TOL, assert_close, winograd_tile
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [
    # name, N, Cin, H, W, Cout, k, stride, pad, dil, bias
    ("1x1_64_256", 2, 64, 33, 29, 256, 1, 1, 0, 1, False),
    ("1x1_256_64", 2, 256, 17, 17, 64, 1, 1, 0, 1, False),
    ("1x1_stride2", 2, 256, 33, 31, 128, 1, 2, 0, 1, False),
    ("3x3_dil1", 1, 64, 35, 35, 64, 3, 1, 1, 1, False),
    ("3x3_dil2", 2, 256, 17, 19, 256, 3, 1, 2, 2, False),
    ("3x3_dil4", 1, 512, 17, 17, 512, 3, 1, 4, 4, False),
    ("aspp_dil12_bias", 2, 128, 33, 33, 256, 3, 1, 12, 12, True),
    ("aspp_dil24_bias", 1, 96, 17, 17, 160, 3, 1, 24, 24, True),   # every off-centre tap falls outside
    ("bottleneck_1280", 1, 1280, 9, 9, 256, 3, 1, 1, 1, True),
    ("stem_7x7", 2, 3, 65, 63, 64, 7, 2, 3, 1, False),
    ("head_19", 2, 256, 17, 17, 19, 1, 1, 0, 1, False),
    ("big_m", 4, 64, 97, 97, 64, 1, 1, 0, 1, False),
    ("wide_ragged_cout", 1, 96, 19, 23, 320, 3, 1, 1, 1, False),    # 256-channel wgrad tile with a partial second tile
    ("wide_stride2", 2, 128, 33, 31, 512, 1, 2, 0, 1, False),         # pixel-index table with a stride
    ("wide_many_splits", 3, 64, 97, 97, 256, 1, 1, 0, 1, False),      # split-K over a ragged pixel range
    ("1x1_ragged_320", 2, 96, 19, 23, 320, 1, 1, 0, 1, False),        # 256 x 256 tile kernel with a partial second column tile
    ("1x1_1024_bias", 1, 256, 33, 31, 1024, 1, 1, 0, 1, True),
]


@pytest.mark.parametrize("tile_cap", [6, 4, 2], ids=["", "F4x4cap", "F2x2cap"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_fwd_bwd(case, tile_cap, monkeypatch):
    """tile_cap 4 / 2 re-run the layers that take larger Winograd tiles by default on F(4x4,3x3) / F(2x2,3x3) (or the direct kernels)
    at those paths' own bounds (conftest.WINO_TOL)."""
    from diga_amd.model import conv as dc
    from diga_amd.model.conv import DigaConv2d
    name, n, cin, h, w, cout, k, stride, pad, dil, bias = case
    default_tile = winograd_tile(n, cin, h, w, cout, k, stride, pad, dil)
    if tile_cap < 6:
        if default_tile <= tile_cap:
            pytest.skip("the default path is this one already")
        monkeypatch.setattr(config.active(), "winograd_max_tile", tile_cap)
    tile = winograd_tile(n, cin, h, w, cout, k, stride, pad, dil)
    # absolute part of the bound, in units of the tensor's scale: y / dx / dw
    a_y, a_dx, a_dw = (WINO_TOL[tile][0], WINO_TOL[tile][0], WINO_TOL[tile][1]) if tile > 2 else (2e-6, 3e-6, 3e-6)
    g = synth.gen(zlib.crc32(name.encode()) % 10000)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    # float64 reference on CPU
    xr = x.double().requires_grad_()
    wr = wt.double().requires_grad_()
    br = b.double().requires_grad_() if bias else None
    yr = F.conv2d(xr, wr, br, stride, pad, dil)
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe.double()).sum().backward()

    m = DigaConv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=bias)
    with torch.no_grad():
        m.weight.copy_(wt)
        if bias:
            m.bias.copy_(b)
    m = m.to(DEV)
    assert m.weight.is_contiguous(memory_format=torch.channels_last)
    need_dx = stride == 1 or k == 1           # the only strided conv with k > 1 is the stem: its input is the image
    xd = x.to(DEV).requires_grad_(need_dx)
    y = m(xd)
    assert tuple(y.shape) == tuple(yr.shape)
    scale = float(yr.abs().max())
    assert_close(y, yr, 1e-5, a_y * scale, f"{name} forward")
    (y * probe.to(DEV)).sum().backward()
    if need_dx:
        gs = float(xr.grad.abs().max())
        assert_close(xd.grad, xr.grad, 1e-5, a_dx * gs, f"{name} grad input")
    ws = float(wr.grad.abs().max())
    assert_close(m.weight.grad, wr.grad, 1e-5, a_dw * ws, f"{name} grad weight")
    assert m.weight.grad.stride() == m.weight.stride()
    if bias:
        assert_close(m.bias.grad, br.grad, 1e-5, 1e-5 * float(br.grad.abs().max()), f"{name} grad bias")


@pytest.fixture
def bf16x3():
    from diga_amd import _lib
    _lib.set_conv_math(1)
    yield
    _lib.set_conv_math(0)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_bf16x3_mode(case, bf16x3):
    """Split-bf16 arithmetic (hi*hi + hi*lo + lo*hi on the bf16 matrix cores) of forward and backward-data:
    errors stay at the 1e-5 level of the output scale, two orders inside the path's 1e-3 logit tolerance."""
    from diga_amd import _lib
    from diga_amd.model.conv import DigaConv2d
    assert _lib.get_conv_math() == 1
    name, n, cin, h, w, cout, k, stride, pad, dil, bias = case
    g = synth.gen(zlib.crc32(name.encode()) % 10000 + 1)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    yr = F.conv2d(xr, wr, None, stride, pad, dil)
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe.double()).sum().backward()
    m = DigaConv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=False)
    with torch.no_grad():
        m.weight.copy_(wt)
    m = m.to(DEV)
    need_dx = stride == 1 or k == 1
    xd = x.to(DEV).requires_grad_(need_dx)
    y = m(xd)
    err = float((y.detach().cpu().double() - yr.detach()).abs().max()) / float(yr.detach().abs().max())
    assert err < 3e-5, f"{name}: forward max error {err:.2e} of the output scale"
    (y * probe.to(DEV)).sum().backward()
    if need_dx:
        gerr = float((xd.grad.cpu().double() - xr.grad).abs().max()) / float(xr.grad.abs().max())
        assert gerr < 3e-5, f"{name}: grad-input max error {gerr:.2e}"
    werr = float((m.weight.grad.cpu().double() - wr.grad).abs().max()) / float(wr.grad.abs().max())
    assert werr < 3e-5, f"{name}: grad-weight max error {werr:.2e}"


def test_model_logits_bf16x3_within_tolerance(golden, bf16x3):
    """north_star tolerance on the whole network in the split-bf16 mode: logits within 1e-3 of the reference."""
    from diga_amd.model.model_noaux import SegModel
    from oracle import deeplab as od
    from oracle import detweights
    g = golden("model")
    m = SegModel()
    m.load_state_dict(detweights.state_dict(od.RESNET101))
    m = m.to(DEV).eval()
    with torch.no_grad():
        out = m(g.t("x").to(DEV))[2]
    scale = float(g.t("out_eval").abs().max())
    assert float((out.cpu() - g.t("out_eval")).abs().max()) < 1e-3 * scale
    m.train()
    m.final.head[0].p = 0.0
    with torch.no_grad():
        out = m(g.t("x").to(DEV))[2]
    scale = float(g.t("out_train").abs().max())
    err = float((out.cpu() - g.t("out_train")).abs().max())
    assert err < 1e-3 * scale, f"train-mode logits off by {err / scale:.2e} of scale"


def test_conv_wgrad_is_deterministic():
    from diga_amd.model.conv import DigaConv2d
    g = synth.gen(3)
    m = DigaConv2d(64, 64, 3, padding=2, dilation=2, bias=False).to(DEV)
    x = torch.randn((2, 64, 65, 65), generator=g).to(DEV)
    grads = []
    for _ in range(2):
        m.weight.grad = None
        m(x).square().sum().backward()
        grads.append(m.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])


def test_conv_rejects_cpu():
    from diga_amd.model.conv import DigaConv2d
    m = DigaConv2d(32, 32, 1)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 32, 4, 4))


@pytest.mark.parametrize("math", [0, 1], ids=["f32", "bf16x3"])
@pytest.mark.parametrize("case", [c for c in CASES if c[5] % 4 == 0 and not c[10]] +
                         [("stats_merge", 2, 32, 192, 193, 64, 1, 1, 0, 1, False),      # > 512 tiles: merge stage
                          ("pw_persist", 4, 64, 193, 193, 256, 1, 1, 0, 1, False)],     # f32: persistent GEMM, 64-row chunks
                         ids=lambda c: c[0])
def test_conv_epilogue_bn_statistics(case, math, monkeypatch):
    """Train-mode BN fed by the per-tile {sum d, sum d^2, shift} partials the conv epilogue emits equals BN that
    re-reads the conv output (and a float64 reference), including running-stat updates; the input carries a large
    mean so a naive sum-of-squares would lose the variance.  (Direct kernels: a Winograd forward emits no partials.)"""
    from diga_amd import _lib
    from diga_amd.model import conv as _dc
    from diga_amd.model.conv import DigaConv2d
    from diga_amd.model.norm import DigaBatchNorm2d
    monkeypatch.setattr(config.active(), "winograd", False)
    name, n, cin, h, w, cout, k, stride, pad, dil, _ = case
    g = synth.gen(zlib.crc32(name.encode()) % 10000 + 7)
    x = torch.randn((n, cin, h, w), generator=g) + 3.0
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    conv = DigaConv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=False)
    bn_a, bn_b = DigaBatchNorm2d(cout), DigaBatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(wt)
        for bn in (bn_a, bn_b):
            bn.weight.copy_(torch.linspace(0.5, 1.5, cout))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, cout))
            for p in bn.parameters():
                p.requires_grad = False
    conv, bn_a, bn_b = conv.to(DEV).train(), bn_a.to(DEV).train(), bn_b.to(DEV).train()
    prev = _lib.get_conv_math()
    _lib.set_conv_math(math)
    try:
        conv.emit_bn_stats = True
        y = conv(x.to(DEV))
        assert hasattr(y, "_diga_bn_partials")
        fused = bn_a(y, relu=True)
        plain = bn_b(y.detach().clone(), relu=True)          # no partials attached: statistics re-read y
    finally:
        _lib.set_conv_math(prev)
    yd = y.detach().double().cpu()
    mean, var = yd.mean((0, 2, 3)), yd.var((0, 2, 3), unbiased=False)
    ref = torch.relu((yd - mean[None, :, None, None]) / torch.sqrt(var + bn_a.eps)[None, :, None, None]
                     * bn_a.weight.double().cpu()[None, :, None, None] + bn_a.bias.double().cpu()[None, :, None, None])
    assert_close(fused.cpu(), ref, rtol=2e-5, atol=2e-5, what=f"{name} fused BN")
    assert_close(fused.cpu(), plain.cpu(), rtol=1e-5, atol=1e-5, what=f"{name} fused vs plain BN")
    cnt = yd.numel() // cout
    assert_close(bn_a.running_mean.cpu(), 0.1 * mean, rtol=1e-5, atol=1e-6, what="running_mean")
    assert_close(bn_a.running_var.cpu(), 0.9 + 0.1 * var * cnt / (cnt - 1), rtol=1e-5, atol=1e-6, what="running_var")


def test_twin_path_matches_register_staged_path_and_is_shared(bf16x3, monkeypatch):
    """The staging-free kernel (pre-split twin + LDS-DMA) computes bit-for-bit what the register-staged kernel does
    (same split, same MFMA order), and convs flagged `share_twin` build the twin of a common input once."""
    from diga_amd import _lib
    from diga_amd.model.conv import DigaConv2d
    g = synth.gen(77)
    x = torch.randn((2, 256, 33, 35), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    convs = [DigaConv2d(256, 160, 3, padding=d, dilation=d, bias=True).to(DEV) for d in (1, 6)] + \
            [DigaConv2d(256, 128, 1, bias=False).to(DEV)]
    with torch.no_grad():
        monkeypatch.setattr(config.active(), "conv_twin", "0")
        ref = [c(x) for c in convs]
        monkeypatch.setattr(config.active(), "conv_twin", "3")
        calls = []
        orig = _lib.call

        def counting(name, *a):
            calls.append(name)
            return orig(name, *a)
        monkeypatch.setattr(_lib, "call", counting)
        for c in convs:
            c.share_twin = True
        got = [c(x) for c in convs]
    assert calls.count("diga_make_twin") == 1 and calls.count("diga_conv2d_nhwc_twin") == 3
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_split_formats_byte_exact():
    """The pre-split operand formats (activation twin, weight LDS images) against their numpy restatement: byte exact,
    including values that round up across a bf16 exponent boundary, denormal-sized residuals and negative zero."""
    import numpy as np
    from diga_amd import _lib
    from oracle import split as osp
    g = synth.gen(123)
    x = torch.randn((37, 64), generator=g) * torch.logspace(-6, 4, 64)[None, :]
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.00390625, 0.99609375, 3.3895314e38, 1e-30])
    xd = x.to(DEV)
    tw = torch.empty(x.numel() * 4, dtype=torch.uint8, device=DEV)
    _lib.call("diga_make_twin", _lib.ptr(xd), 64, _lib.ptr(tw), 37, 64, _lib.stream())
    assert np.array_equal(tw.cpu().numpy(), osp.twin(x.numpy()))
    for k in (19 * 4, 160):                                   # one 128-row tile with padding rows / two tiles
        w = torch.randn((k, 9, 64), generator=g) * 0.05
        img = torch.empty(_lib.lib.diga_split_bf16_image_bytes(k, 9, 64), dtype=torch.uint8, device=DEV)
        _lib.call("diga_split_bf16_image", _lib.ptr(w.to(DEV)), _lib.ptr(img), k, 9, 64, _lib.stream())
        assert np.array_equal(img.cpu().numpy(), osp.weight_image(w.numpy()))


@pytest.mark.parametrize("geom", [(2, 64, 96, 33, 35, 3, 2), (1, 256, 320, 19, 23, 3, 1), (3, 128, 256, 40, 40, 1, 1),
                                  (2, 96, 264, 17, 17, 3, 6)], ids=["3x3d2", "ragged_cout", "1x1", "dil6_outside"])
def test_wgrad_on_twins_vs_float64(geom, bf16x3):
    """Backward-weight from the split twins (LDS-DMA staging, transposing LDS reads) against a float64 reference."""
    from diga_amd import _lib
    n, cin, cout, h, w, k, dil = geom
    g = synth.gen(sum(geom))
    pad = dil * (k // 2)
    x = torch.randn((n, h, w, cin), generator=g)
    dy = torch.randn((n, h, w, cout), generator=g)
    xr = x.permute(0, 3, 1, 2).double()
    wr = torch.zeros((cout, cin, k, k), dtype=torch.float64, requires_grad=True)
    (F.conv2d(xr, wr, None, 1, pad, dil) * dy.permute(0, 3, 1, 2).double()).sum().backward()
    want = wr.grad.permute(0, 2, 3, 1).contiguous()                      # [K][R][S][C]
    xd, dyd = x.to(DEV), dy.to(DEV)
    m = n * h * w
    xt = torch.empty(m * cin * 4, dtype=torch.uint8, device=DEV)
    dyt = torch.empty(m * cout * 4, dtype=torch.uint8, device=DEV)
    _lib.call("diga_make_twin", _lib.ptr(xd), cin, _lib.ptr(xt), m, cin, _lib.stream())
    _lib.call("diga_make_twin", _lib.ptr(dyd), cout, _lib.ptr(dyt), m, cout, _lib.stream())
    dw = torch.empty((cout, k, k, cin), dtype=torch.float32, device=DEV)
    ws = torch.empty(_lib.lib.diga_conv2d_wgrad_twin_workspace_bytes(n, h, w, cout, cin, k, k), dtype=torch.uint8, device=DEV)
    for _ in range(2):
        _lib.call("diga_conv2d_wgrad_twin", _lib.ptr(dyt), _lib.ptr(xt), _lib.ptr(dw), _lib.ptr(ws), ws.numel(), n, h, w, cin, h, w,
                  cout, k, k, 1, 1, -pad, -pad, dil, dil, _lib.stream())
    scale = float(want.abs().max())
    assert_close(dw.cpu(), want, rtol=0.0, atol=5e-5 * scale, what="dw from twins")


def _conv_pair(cin, cout, k, s, p, d, wt, rows=None, cols=None):
    """DigaConv2d on the device holding wt[rows, cols] (a slice of the full layer's weight)."""
    from diga_amd.model.conv import DigaConv2d
    w = wt if rows is None else wt[rows]
    w = w if cols is None else w[:, cols]
    m = DigaConv2d(w.shape[1], w.shape[0], k, stride=s, padding=p, dilation=d, bias=False).to(DEV)
    with torch.no_grad():
        m.weight.copy_(w)
    return m


DMA_EQ_CASES = [("aspp_d24", 1, 256, 97, 97, 256, 3, 1, 24, 24), ("pw_tail", 2, 256, 37, 41, 320, 1, 1, 0, 1),
                ("pw_big", 3, 256, 193, 161, 256, 1, 1, 0, 1), ("stride2", 2, 256, 65, 65, 256, 1, 2, 0, 1),
                ("d2_3x3", 3, 256, 33, 29, 256, 3, 1, 2, 2), ("d12", 1, 288, 97, 97, 256, 3, 1, 12, 12)]


@pytest.mark.parametrize("case", DMA_EQ_CASES, ids=[c[0] for c in DMA_EQ_CASES])
def test_f32_dma_and_persistent_kernels_bit_identical_to_register_staged_kernel(case, monkeypatch):
    """conv_fwd_dma_kernel (256 x 128 tiles, LDS-DMA operands, dead taps skipped) and gemm_f32_persistent_kernel (the same tile
    walked by 256 persistent blocks) run K in conv_fwd_kernel's order with the same MFMA per (k-group, element): outputs and
    input gradients are equal BIT FOR BIT.  No switch is involved -- the dispatch is by shape, so the same layer is evaluated
    twice: whole (>= 256 output channels from K >= 256: the LDS-DMA kernel; pw_big's 730 tiles: the persistent GEMM) and as two
    layers on halves of its output channels (< 256 wide: the register-staged kernel); likewise the input gradient, whole and for
    halves of the input channels (a backward-data convolution into < 256 channels).  Direct kernels only (Winograd off)."""
    from diga_amd import _lib
    from diga_amd.model import conv as dc
    monkeypatch.setattr(config.active(), "winograd", False)
    name, n, cin, h, w, cout, k, s, p, d = case
    g = torch.Generator().manual_seed(4321)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * 0.05
    prev = _lib.get_conv_math()
    _lib.set_conv_math(0)
    calls, real = [], _lib.call
    try:
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
        y = _conv_pair(cin, cout, k, s, p, d, wt)(xd)
        probe = torch.randn(y.shape, generator=g).to(DEV)
        (y * probe).sum().backward()
        half = 128
        for r0, r1 in ((0, half), (half, cout)):                 # forward: halves of the output channels
            ys = _conv_pair(cin, cout, k, s, p, d, wt, rows=slice(r0, r1))(xd.detach())
            assert torch.equal(ys, y[:, r0:r1]), (name, "y", r0)
        for c0, c1 in ((0, half), (half, cin)):                  # backward-data: halves of the input channels
            xs = x[:, c0:c1].to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
            monkeypatch.setattr(_lib, "call", lambda fn, *a: (calls.append(fn), real(fn, *a))[1])
            (_conv_pair(cin, cout, k, s, p, d, wt, cols=slice(c0, c1))(xs) * probe).sum().backward()
            monkeypatch.setattr(_lib, "call", real)
            assert torch.equal(xs.grad, xd.grad[:, c0:c1]), (name, "dx", c0)
    finally:
        _lib.set_conv_math(prev)
    assert not any("winograd" in c for c in calls), calls


WINO_CASES = [("d1_ragged", 3, 128, 33, 29, 160, 1), ("d1_wide", 2, 384, 31, 37, 512, 1), ("d18_97", 1, 128, 97, 97, 256, 18), ("d2_97", 2, 256, 97, 97, 256, 2), ("d4_65x129", 1, 512, 65, 129, 128, 4),
              ("d6_97", 1, 256, 97, 97, 256, 6), ("d12_97", 1, 128, 97, 97, 128, 12), ("d24_97", 1, 128, 97, 97, 132, 24),
              ("d3_tiny", 2, 128, 5, 7, 128, 3)]


@pytest.mark.parametrize("tile", [4, 6], ids=["F4x4", "F6x6"])
@pytest.mark.parametrize("case", WINO_CASES + [("d2_4img", 4, 128, 49, 53, 128, 2), ("d1_k512", 2, 128, 40, 40, 512, 1)], ids=lambda c: c[0])
def test_winograd_output_transform_bn_statistics(case, tile, monkeypatch):
    """Round 5: the forward Winograd output transform (4x4 / 6x6 tiles) leaves the BatchNorm behind the layer its column statistics as
    RECORDS of unequal size (tile groups hold different numbers of in-image pixels; diga_bn_fwd_records) -- the BatchNorm's own
    statistics pass over y is gone.  Train-mode BN fed by the records equals BN that re-reads the conv output and a float64
    reference, running statistics included; the input carries a large mean; the conv output itself is bit-identical either way."""
    from diga_amd.model import conv as dc
    from diga_amd.model.conv import DigaConv2d
    from diga_amd.model.norm import DigaBatchNorm2d
    name, n, cin, h, w, cout, dil = case
    if cout % 4 != 0:
        pytest.skip("statistics need Cout % 4 == 0")
    monkeypatch.setattr(config.active(), "winograd_max_tile", tile)
    monkeypatch.setattr(config.active(), "winograd_ratio", 10.0)
    if winograd_tile(n, cin, h, w, cout, 3, 1, dil, dil) != tile:
        pytest.skip("this geometry does not take this tile")
    g = synth.gen(zlib.crc32(name.encode()) % 10000 + 11 + tile)
    x = torch.randn((n, cin, h, w), generator=g) + 2.0
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    conv = DigaConv2d(cin, cout, 3, stride=1, padding=dil, dilation=dil, bias=False)
    bn_a, bn_b = DigaBatchNorm2d(cout), DigaBatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(wt)
        for bn in (bn_a, bn_b):
            bn.weight.copy_(torch.linspace(0.5, 1.5, cout))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, cout))
            for p in bn.parameters():
                p.requires_grad = False
    conv, bn_a, bn_b = conv.to(DEV).train(), bn_a.to(DEV).train(), bn_b.to(DEV).train()
    conv.emit_bn_stats = True
    xd = x.to(DEV)
    y = conv(xd)
    part = getattr(y, "_diga_bn_partials", None)
    assert part is not None and part[1][0] == "records" and part[1][1] > 0
    fused = bn_a(y, relu=True)
    monkeypatch.setattr(config.active(), "winograd_stats", False)
    y2 = conv(xd)
    assert not hasattr(y2, "_diga_bn_partials") and torch.equal(y2, y)
    plain = bn_b(y2, relu=True)
    yd = y.detach().double().cpu()
    mean, var = yd.mean((0, 2, 3)), yd.var((0, 2, 3), unbiased=False)
    ref = torch.relu((yd - mean[None, :, None, None]) / torch.sqrt(var + bn_a.eps)[None, :, None, None]
                     * bn_a.weight.double().cpu()[None, :, None, None] + bn_a.bias.double().cpu()[None, :, None, None])
    assert_close(fused.cpu(), ref, rtol=2e-5, atol=2e-5, what=f"{name} BN on records")
    assert_close(fused.cpu(), plain.cpu(), rtol=1e-5, atol=1e-5, what=f"{name} records vs statistics pass")
    cnt = yd.numel() // cout
    assert_close(bn_a.running_mean.cpu(), 0.1 * mean, rtol=1e-5, atol=1e-6, what="running_mean")
    assert_close(bn_a.running_var.cpu(), 0.9 + 0.1 * var * cnt / (cnt - 1), rtol=1e-5, atol=1e-6, what="running_var")
    # the counts of the records add up to the layer's pixels
    recs = int(part[1][1])
    counts = part[0][recs * 3 * cout: recs * 3 * cout + recs]
    assert float(counts.sum()) == n * h * w


@pytest.mark.parametrize("tile", [2, 4, 6], ids=["F2x2", "F4x4", "F6x6"])
@pytest.mark.parametrize("case", WINO_CASES, ids=[c[0] for c in WINO_CASES])
def test_winograd_f32_vs_float64(case, tile, monkeypatch):
    """Winograd F(2x2,3x3) and F(4x4,3x3) (csrc/winograd.hip: sub-image tiling of the dilated conv, input / weight / output
    transforms around one batched launch of the fp32 LDS-DMA GEMM) against torch's float64 CPU convolution, forward,
    backward-data (the flipped-tap call) and backward-weight; odd map sizes, images whose sub-images are smaller than a tile, a
    Cout tail, bias.  Bounds, relative to the tensor's scale: F(2x2) 1e-5 = what the direct fp32 kernels are held to; F(4x4)
    3e-5, weight gradient 5e-5 (its transforms carry coefficients up to 8 and 1/24: measured <= 1.7e-5 / 2.9e-5).  The multiplication-ratio
    gate is opened and the tile forced so that every case takes the path under test."""
    from diga_amd import _lib
    from diga_amd.model import conv as dc
    name, n, cin, h, w, cout, d = case
    monkeypatch.setattr(config.active(), "winograd", True)
    monkeypatch.setattr(config.active(), "winograd_ratio", 10.0)
    monkeypatch.setattr(dc, "_wino_plan", lambda hi, wi, dd: (tile, 0.5))
    calls = []
    real = _lib.call

    def spy(fname, *a):
        calls.append(fname)
        return real(fname, *a)
    monkeypatch.setattr(_lib, "call", spy)
    g = synth.gen(zlib.crc32(name.encode()) % 10000 + 31)
    x = torch.randn((n, cin, h, w), generator=g) + 0.5
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g)
    xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), b.double().requires_grad_()
    yr = F.conv2d(xr, wr, br, 1, d, d)
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe.double()).sum().backward()
    m = dc.DigaConv2d(cin, cout, 3, stride=1, padding=d, dilation=d, bias=True)
    with torch.no_grad():
        m.weight.copy_(wt)
        m.bias.copy_(b)
    m = m.to(DEV)
    prev = _lib.get_conv_math()
    _lib.set_conv_math(0)
    try:
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
        y = m(xd)
        (y * probe.to(DEV)).sum().backward()
    finally:
        _lib.set_conv_math(prev)
    wide = cout % 256 == 0 and cin % 128 == 0          # weight gradient through Winograd too, on the V the forward kept
    assert calls.count("diga_conv2d_winograd_f32") + calls.count("diga_conv2d_winograd_f32_keep") == 2, calls   # forward + backward-data
    assert ("diga_conv2d_wgrad_winograd_f32" in calls) == wide and ("diga_conv2d_winograd_f32_keep" in calls) == wide, calls
    for got, want, what in ((y, yr, "y"), (xd.grad, xr.grad, "dx"), (m.weight.grad, wr.grad, "dw")):
        e = float((got.detach().cpu().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert e < WINO_TOL[tile][1 if what == "dw" else 0], (what, e)
        print(f"winograd tile {tile} {name} {what}: max err / scale = {e:.2e}")
    if wide:                                           # ... and the same gradient when the backward recomputes V
        monkeypatch.setattr(config.active(), "winograd_keep_v", False)
        dw_kept = m.weight.grad.clone()
        m.weight.grad = None
        _lib.set_conv_math(0)
        try:
            (m(xd.detach()) * probe.to(DEV)).sum().backward()
        finally:
            _lib.set_conv_math(prev)
        assert torch.equal(m.weight.grad, dw_kept)


@pytest.mark.parametrize("tile", [2, 4, 6], ids=["F2x2", "F4x4", "F6x6"])
def test_persistent_gemm_bit_identical_to_per_tile_launch(tile, monkeypatch):
    """The Winograd products on gemm_f32_persistent_kernel (256 blocks walking the tiles, next tile's stages prefetched under
    the current one, accumulators stored from registers) equal the per-tile launch of conv_fwd_dma_kernel bit for bit.  By shape,
    without a switch: three 97x97 images in one call make >= 512 GEMM tiles (the persistent walk), each image alone stays
    below (one block per tile) -- and a tile's products do not depend on which other tiles are in the launch."""
    from diga_amd import _lib
    from diga_amd.model import conv as dc
    from diga_amd.model.conv import DigaConv2d
    monkeypatch.setattr(config.active(), "winograd", True)
    monkeypatch.setattr(config.active(), "winograd_ratio", 10.0)
    monkeypatch.setattr(dc, "_wino_plan", lambda hi, wi, dd: (tile, 0.5))
    prev = _lib.get_conv_math()
    _lib.set_conv_math(0)
    try:
        torch.manual_seed(5)
        g = torch.Generator().manual_seed(99)
        for name, n, cin, h, w, cout, d in [("d2", 3, 128, 97, 97, 256, 2), ("d6", 3, 256, 97, 97, 256, 6)]:
            prod = (tile + 2) ** 2
            tiles_1 = sum(-(-((h - a + d - 1) // d) // tile) for a in range(d)) ** 2
            grid = lambda imgs: prod * (-(-imgs * tiles_1 // 256) * 256) // 256 * (cout // 128)          # noqa: E731
            assert grid(n) >= 512 > grid(1), (grid(n), grid(1))
            x = torch.randn((n, cin, h, w), generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
            m = DigaConv2d(cin, cout, 3, padding=d, dilation=d, bias=False).to(DEV)
            y = m(x)
            probe = torch.randn(y.shape, generator=g).to(DEV)
            (y * probe).sum().backward()
            for i in range(n):
                xi = x[i:i + 1].detach().clone(memory_format=torch.channels_last).requires_grad_()
                yi = m(xi)
                (yi * probe[i:i + 1]).sum().backward()
                assert torch.equal(yi, y[i:i + 1]), (name, "y", i)
                assert torch.equal(xi.grad, x.grad[i:i + 1]), (name, "dx", i)
    finally:
        _lib.set_conv_math(prev)
