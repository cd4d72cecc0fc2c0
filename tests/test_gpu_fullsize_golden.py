"""The benchmark geometries pinned against the REFERENCE (round 3): captures of the reference `SegModel`
(G5/model/model_noaux.py:28-46, seg_model_noaux.py:200-214) in train mode on two crops of 768x768 (97x97 map,
BASELINE configs[1]) and of 512x1024 (65x129 map, configs[3]) -- tests/golden/full768.npz, full512x1024.npz, made
by tools/gen_golden.py from the imported reference -- against the HIP path in both conv arithmetics; plus one
stand-alone convolution per ASPP dilation on a 97x97 map (Cin >= 256: twin + dead-tap path) against a float64 CPU
convolution.  Until this file every full-size check compared the build with itself."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle import deeplab as od
from oracle import detweights, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model():
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    m = SegModel(arch=sm.RESNET101)
    m.load_state_dict(detweights.state_dict(od.RESNET101))
    return m.to(DEV)


@pytest.fixture(params=[False, True], ids=["tiles6", "exact_tiles2"])
def exact_tiles(request, conv_math):
    """exact_tiles2: `_lib.set_conv_math(0, exact=True)` -- every Winograd layer on F(2x2,3x3), the direct fmaf chain's error level;
    held to the bounds the path had before the larger tiles (round 3), so that a regression of the direct / F(2x2) kernels still shows."""
    from diga_amd import _lib
    if request.param and conv_math != 0:
        pytest.skip("the exact setting belongs to the fp32 arithmetic")
    if request.param:
        _lib.set_conv_math(0, exact=True)
    yield request.param
    if request.param:
        _lib.set_conv_math(0, exact=False)


@pytest.mark.parametrize("name", ["full768", "full512x1024", "full768b8", "full512x1024b8"])
def test_benchmark_geometry_vs_reference(golden, conv_math, exact_tiles, name):
    """full768b8 (round 5): EIGHT images of 768 x 768 -- half the benchmark's student batch, 75 272 rows per pointwise GEMM (beyond 2^15
    and 2^16 rows; four times the tile count, other split-K boundaries than the two-image captures); its logits are stored as a
    strided sample.  full512x1024b8: the self-training geometry at the per-GPU batch of its student(cat) pass (4 + 4 crops, 67 080 rows)."""
    from diga_amd import _lib
    g = golden(name)
    batch, H, W = (int(v) for v in g["geometry"])
    gen = synth.gen(int(g["seed"]))
    x = torch.rand((batch, 3, H, W), generator=gen) * 2 - 1
    m = _model().train()
    m.final.head[0].p = 0.0                                       # Dropout2d off, as in the capture
    sh, dp, out, feat = m(x.to(DEV))
    if "out" in g:
        want = g.t("out")
        assert tuple(out.shape) == tuple(want.shape)
        got = out.detach().cpu()
    else:
        want = g.t("out_sample")
        got = out.detach().cpu().reshape(-1)[::7]
        assert tuple(got.shape) == tuple(want.shape) and tuple(out.shape[:2]) == (batch, 19)
        assert float(out.detach().abs().sum()) == pytest.approx(float(g["out_sum"][1]), rel=1e-4)
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    # north_star: logits within 1e-3 relative of the reference's CPU path -- in BOTH arithmetics ...
    assert err < 1e-3 * scale, (name, err, scale)
    # ... and within 3x of what the kernels measure (exact fp32 with 6x6 / 4x4 Winograd tiles: 2.5e-5 / 4.3e-5 of scale at 768x768 /
    # 512x1024; split bf16: 1.6e-4): a regression of the arithmetic shows long before it reaches the contract's bound
    assert err < ((4e-5 if exact_tiles else 1.5e-4) if conv_math == 0 else 5e-4) * scale, (name, err, scale)
    if conv_math == 0:
        assert_close(got, want, 1e-3, 3e-4 * scale, "train logits (fp32 mode, elementwise)")
    # features / trunk outputs: strided samples and L1 sums of the reference
    featc = feat.detach().cpu()
    fs = g.t("feat_sample")
    assert float((featc.reshape(-1)[::61] - fs).abs().max()) < 1e-3 * float(fs.abs().max())
    assert float(featc.abs().sum()) == pytest.approx(float(g["feat_sum"][1]), rel=1e-4)
    dpc = dp.detach().cpu()
    ds = g.t("deep_sample")
    assert float((dpc.reshape(-1)[::9973] - ds).abs().max()) < 1e-3 * float(ds.abs().max())
    assert float(dpc.abs().sum()) == pytest.approx(float(g["deep_sum"][1]), rel=1e-4)
    assert float(sh.detach().abs().sum()) == pytest.approx(float(g["shallow_sum"][1]), rel=1e-4)
    del featc, dpc

    probe = torch.randn(tuple(out.shape), generator=gen)
    _lib.side_overlap = True                                      # as bench.py runs it: weight gradients on the side stream
    try:
        (out * probe.to(DEV)).sum().backward()
    finally:
        _lib.side_overlap = False
        _lib.join_side()
    named = dict(m.named_parameters())
    ref = g.t("g_head")
    assert_close(named["final.head.1.weight"].grad, ref, 5e-3, 2e-3 * float(ref.abs().max()), "head gradient")
    # Weight gradients of 22 layers (every stride / downsample transition, all four ASPP dilations, SE, GN).  Measured
    # (tools/diag/fullsize_grad_errors.py, both geometries):
    #   * norms (L1, L2) agree to <= 2e-4 in fp32 and <= 9e-4 in bf16x3 everywhere;
    #   * layers whose gradient does not pass back through a trunk ReLU (bottleneck conv / GN / SE, the branch GroupNorms)
    #     agree elementwise to 1e-5 (fp32) / 2e-4 (bf16x3) of scale;
    #   * every layer below a ReLU deviates by a uniform 2e-3..6e-3 (fp32, F(2x2) Winograd: 7e-6 forward differences) /
    #     9e-3..1.2e-2 (fp32 with the F(6x6) / F(4x4) tiles of round 4: 2.5e-5..4e-5 forward differences) / 1e-2..3e-2 (bf16x3:
    #     1.6e-4) in relative L2, largest elements 2e-2 / 4e-2: the gradients are discontinuous where a pre-activation crosses zero
    #     (a 2e-7 weight perturbation moves them by 4e-3 in L2 on the CPU reference itself), and 1e-6 / 1e-5-level
    #     forward differences flip a few of the 10^9 ReLUs.  An indexing error moves norms and samples by O(1).
    fp32 = conv_math == 0
    tol_norm = 1e-3 if fp32 else 3e-3
    tol_l2, tol_max = ((1.2e-2, 4e-2) if exact_tiles else (2.5e-2, 6e-2)) if fp32 else (5e-2, 8e-2)
    tol_smooth = 1e-4 if fp32 else 1e-3
    smooth = ("final_bottleneck_0_se_0_weight", "final_bottleneck_1_bias", "final_bottleneck_1_weight", "final_bottleneck_2_weight")
    keys = sorted(k[2:-5] for k in g if k.startswith("g_") and k.endswith("__sum"))
    assert len(keys) == 22
    by_flat = {n.replace(".", "_"): n for n in named}
    worst = 0.0
    for k in keys:
        gr = named[by_flat[k]].grad.detach().cpu()
        _, l1, l2 = (float(v) for v in g["g_" + k + "__sum"])
        assert float(gr.abs().sum()) == pytest.approx(l1, rel=tol_norm), k
        assert float(gr.norm()) == pytest.approx(l2, rel=tol_norm), k
        step = int(g["g_" + k + "__step"])
        smp = g.t("g_" + k + "__sample")
        d = gr.reshape(-1)[::step] - smp
        e_max, e_l2 = float(d.abs().max() / smp.abs().max()), float(d.norm() / smp.norm())
        worst = max(worst, e_l2)
        if k in smooth:
            assert e_max < tol_smooth, (k, e_max)
        else:
            assert e_l2 < tol_l2 and e_max < tol_max, (k, e_l2, e_max)
    sd = m.state_dict()
    assert_close(sd["layer1.0.bn1.running_mean"], g.t("rm_layer1"), 1e-4, 1e-6, "running mean layer1")
    assert_close(sd["layer3.22.bn3.running_mean"], g.t("rm_layer3"), 1e-3, 1e-5, "running mean layer3")
    assert_close(sd["layer4.2.bn3.running_var"], g.t("rv_layer4"), 1e-3, 1e-6, "running var layer4")
    # eval mode (running statistics as updated by the pass above, like the capture)
    if "out_eval" in g:
        m.eval()
        with torch.no_grad():
            oe = m(x.to(DEV))[2]
        we = g.t("out_eval")
        assert float((oe.cpu() - we).abs().max()) < 1e-3 * float(we.abs().max())
    print(f"{name} math={conv_math} exact={exact_tiles}: logits max err {err / scale:.2e} of scale, worst gradient-sample relative L2 {worst:.2e}")


ASPP_CASES = [("d6", 6, 256), ("d12", 12, 256), ("d18", 18, 288), ("d24", 24, 256), ("d24_wide", 24, 512)]


@pytest.mark.parametrize("case", ASPP_CASES, ids=[c[0] for c in ASPP_CASES])
def test_aspp_dilations_at_97x97_vs_float64(conv_math, case):
    """ASPP-shaped 3x3 convolutions (Cin -> 256, +bias, dilation = padding = d) on the 97x97 map of the benchmark: the
    twin kernel with dead-tap skipping on a map where off-centre taps of dilation 24 are live for most rows; forward,
    backward-data and backward-weight against torch's float64 CPU convolution."""
    from diga_amd.model.conv import DigaConv2d
    _, d, cin = case
    g = synth.gen(9700 + d + cin)
    x = torch.randn((1, cin, 97, 97), generator=g)
    wt = torch.randn((256, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(256, generator=g)
    xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), b.double().requires_grad_()
    yr = F.conv2d(xr, wr, br, 1, d, d)
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe.double()).sum().backward()
    m = DigaConv2d(cin, 256, 3, stride=1, padding=d, dilation=d, bias=True)
    with torch.no_grad():
        m.weight.copy_(wt)
        m.bias.copy_(b)
    m = m.to(DEV)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = m(xd)
    (y * probe.to(DEV)).sum().backward()
    # of the output scale; exact fp32: per Winograd tile (d = 6 / 18: 6x6, d = 12 / 24: 4x4 tiles), conftest.WINO_TOL
    from conftest import WINO_TOL, winograd_tile
    tile = winograd_tile(1, cin, 97, 97, 256, 3, 1, d, d) if conv_math == 0 else 0
    f4 = tile > 2
    tol = WINO_TOL[tile][0] if conv_math == 0 else 3e-5
    for got, want, what in ((y, yr, "y"), (xd.grad, xr.grad, "dx"), (m.weight.grad, wr.grad, "dw"), (m.bias.grad, br.grad, "db")):
        e = float((got.detach().cpu().double() - want.detach()).abs().max() / want.detach().abs().max())
        assert e < (WINO_TOL[tile][1] if f4 and what == "dw" else tol), (what, e)
        print(f"aspp {case[0]} math={conv_math} {what}: {e:.2e} of scale")
