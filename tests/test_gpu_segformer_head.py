"""The SegFormer decode head on the HIP kernels (diga_amd/model/networks/segformer_head.py: fuse conv folded into the per-stage
embeddings, diga_pyramid_sum_fwd/_bwd, diga_bn_bwd_affine) against the capture of the REFERENCE class
(tests/golden/segformer_head.npz, G5/model/networks/segformer_head.py:25-165) and against the oracle on other geometries."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle import segformer_head as oh
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
CHANS = [64, 128, 320, 512]
# measured against the capture of the reference class (tools/diag run, both conv arithmetics), in units of each tensor's scale:
#   exact fp32 : logits 1.2e-6, input gradients 1.2e-6, parameter gradients 1.2e-6
#   split bf16 : logits 8.8e-6, input gradients 8.2e-6, parameter gradients 9.6e-6
# the bounds are 3-4x that, per arithmetic
TOLS = {0: {"logits": 4e-6, "dinput": 4e-6, "dparam": 5e-6}, 1: {"logits": 3e-5, "dinput": 3e-5, "dparam": 3e-5}}
TOL = TOLS[1]                 # (tests that do not take the conv_math fixture run in whatever mode the process is in)


def _head(embed=768, return_raw=False, sd=None, classes=19):
    from diga_amd.model.networks.segformer_head import SegFormerHead
    h = SegFormerHead(in_channels=CHANS, channels=128, feature_strides=[4, 8, 16, 32], num_classes=classes, in_index=[0, 1, 2, 3],
                      dropout_ratio=0.1, align_corners=False, decoder_params={"embed_dim": embed}, return_raw=return_raw)
    h.load_state_dict(sd if sd is not None else oh.state_dict(CHANS, classes, embed))
    h.dropout.p = 0.0
    return h.to(DEV)


def test_state_dict_keys_are_the_references(golden):
    from diga_amd.model.networks.segformer_head import SegFormerHead
    h = SegFormerHead(in_channels=CHANS, channels=128, feature_strides=[4, 8, 16, 32], num_classes=19, in_index=[0, 1, 2, 3])
    assert list(h.state_dict().keys()) == golden("segformer_head")["keys"].tolist()


def test_train_step_vs_reference_capture(golden, conv_math):
    TOL = TOLS[conv_math]
    g = golden("segformer_head")
    h = _head().train()
    feats = [g.t(f"c{i}").to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_() for i in (1, 2, 3, 4)]
    logits, feat = h(feats)
    want = g.t("logits")
    assert tuple(logits.shape) == tuple(want.shape) and tuple(feat.shape) == (2, 768, 32, 24)
    err = float((logits.detach().cpu() - want).abs().max() / want.abs().max())
    assert err < TOL["logits"], err
    (logits * g.t("probe").to(DEV)).sum().backward()
    for i, f in enumerate(feats):
        w = g.t(f"dc{i + 1}")
        e = float((f.grad.cpu() - w).abs().max() / w.abs().max())
        assert e < TOL["dinput"], (i, e)
    named = dict(h.named_parameters())
    for k in [n[2:] for n in g if n.startswith("g_")]:
        name = [n for n in named if n.replace(".", "_") == k][0]
        step = int(g["gstep_" + k])
        w = g.t("g_" + k)
        got = named[name].grad.detach().cpu().reshape(-1)[::step]
        if name.endswith("proj.bias"):
            assert float(got.abs().max()) < 5e-5, name        # null direction of a train-mode BatchNorm: rounding noise in both
            continue
        e = float((got - w).abs().max() / w.abs().max())
        assert e < TOL["dparam"], (name, e)
        # (whole-tensor norms in float64 on both sides: torch's fp32 norm of the 2.4 M-element fuse weight is off by 5e-5)
        assert abs(float(named[name].grad.double().norm()) / float(g["gnorm_" + k]) - 1) < TOL["dparam"], name
    assert_close(h.linear_fuse.bn.running_mean, g.t("running_mean"), 1e-4, 1e-5, "running_mean")
    assert_close(h.linear_fuse.bn.running_var, g.t("running_var"), 1e-4, 1e-5, "running_var")
    assert int(h.linear_fuse.bn.num_batches_tracked) == 1


def test_eval_mode_odd_sizes_vs_reference_capture(golden, conv_math):
    TOL = TOLS[conv_math]
    g = golden("segformer_head")
    sd = oh.state_dict()
    sd["linear_fuse.bn.running_mean"], sd["linear_fuse.bn.running_var"] = g.t("running_mean"), g.t("running_var")
    h = _head(sd=sd).eval()
    feats = [g.t(f"e{i}").to(DEV) for i in (1, 2, 3, 4)]
    with torch.no_grad():
        logits, _ = h(feats)
    want = g.t("logits_eval")
    assert tuple(logits.shape) == (1, 19, 25, 19)
    err = float((logits.cpu() - want).abs().max() / want.abs().max())
    assert err < TOL["logits"], err


def test_raw_features_second_output(golden):
    g = golden("segformer_head")
    h = _head(return_raw=True).train()
    feats = [g.t(f"c{i}").to(DEV) for i in (1, 2, 3, 4)]
    with torch.no_grad():
        logits, c_raw = h(feats)
    assert tuple(c_raw.shape) == (2, 3072, 32, 24)
    assert float((c_raw.cpu().reshape(-1)[::397] - g.t("c_raw_sample")).abs().max()) < 2e-5
    assert abs(synth.checksum(c_raw.cpu()) - float(g["c_raw_sum"])) < 1e-4 * abs(float(g["c_raw_sum"])) + 1e-2


@pytest.mark.parametrize("geom", [((25, 19), [(13, 10), (7, 5), (4, 3)]), ((64, 48), [(8, 6)]), ((17, 40), [(17, 40), (3, 9)]),
                                  ((12, 12), [(30, 5), (1, 1)])])
def test_pyramid_sum_kernels_vs_torch(geom):
    """diga_pyramid_sum_fwd / _bwd against F.interpolate(mode='bilinear', align_corners=False) and its autograd: integer and
    fractional ratios, a same-size source, a source LARGER than the target along one axis, a 1x1 source."""
    from diga_amd.model.networks.segformer_head import _PyramidSumFn
    (H, W), coarse = geom
    gen = synth.gen(H * 100 + W)
    n, c = 2, 24
    fine = torch.randn((n, c, H, W), generator=gen)
    bias = torch.randn(c, generator=gen)
    srcs = [torch.randn((n, c, h, w), generator=gen) for h, w in coarse]
    probe = torch.randn((n, c, H, W), generator=gen)
    ref_in = [t.clone().double().requires_grad_() for t in [fine, bias] + srcs]
    ref = ref_in[0] + ref_in[1][None, :, None, None] + sum(F.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)
                                                            for t in ref_in[2:])
    (ref * probe.double()).sum().backward()
    dev_in = [t.to(DEV).requires_grad_() for t in [fine, bias] + srcs]
    x = dev_in[0].contiguous(memory_format=torch.channels_last) * 1.0        # a fresh non-leaf buffer the op may overwrite
    out = _PyramidSumFn.apply(x, dev_in[1], *dev_in[2:])
    (out * probe.to(DEV)).sum().backward()
    assert_close(out, ref, 2e-6, 1e-5, "pyramid sum")
    for got, want, what in zip(dev_in, ref_in, ["fine", "bias"] + [f"src{k}" for k in range(len(srcs))]):
        assert_close(got.grad, want.grad, 1e-5, 1e-5 * float(want.grad.abs().max()), "grad " + what)


@pytest.mark.parametrize("n,c,H,W", [(2, 32, 48, 64), (1, 16, 16, 16), (3, 48, 32, 80)])
def test_fused_three_ratio_backward_vs_per_ratio_kernels(n, c, H, W):
    """diga_pyramid_sum_bwd3 (LDS-staged tile, all of ratios 2 / 4 / 8 from one pass over the fine gradient) against the three
    per-ratio gathers: same weights, another summation order."""
    from diga_amd import _lib
    gen = synth.gen(H + W + c)
    g = torch.randn((n, H, W, c), generator=gen).to(DEV)
    outs3 = [torch.empty((n, H // r, W // r, c), device=DEV) for r in (2, 4, 8)]
    _lib.call("diga_pyramid_sum_bwd3", _lib.ptr(g), H, W, *[_lib.ptr(o) for o in outs3], n, c, _lib.stream())
    for r, o3 in zip((2, 4, 8), outs3):
        o1 = torch.empty_like(o3)
        _lib.call("diga_pyramid_sum_bwd", _lib.ptr(g), H, W, _lib.ptr(o1), H // r, W // r, n, c, _lib.stream())
        assert_close(o3, o1, 2e-6, 2e-6 * float(o1.abs().max()), f"ratio {r}")


@pytest.mark.parametrize("n,c,H,W", [(2, 64, 48, 64), (1, 32, 16, 16), (3, 96, 32, 80)])
def test_tiled_forward_bit_identical_and_its_batchnorm_statistics(n, c, H, W):
    """diga_pyramid_sum_fwd3 (coarse tiles staged in LDS) must reproduce the flat kernel to the last bits, and the per-chunk statistics it
    leaves for the BatchNorm must finalise to the mean / variance of the map (float64 reference)."""
    from diga_amd import _lib
    gen = synth.gen(H * 3 + W + c)
    fine = torch.randn((n, H, W, c), generator=gen).to(DEV)
    bias = torch.randn(c, generator=gen).to(DEV)
    srcs = [torch.randn((n, H // r, W // r, c), generator=gen).to(DEV) for r in (2, 4, 8)]
    a, b = fine.clone(), fine.clone()
    _lib.call("diga_pyramid_sum_fwd", _lib.ptr(a), H, W, _lib.ptr(bias), _lib.ptr(srcs[0]), H // 2, W // 2, _lib.ptr(srcs[1]), H // 4, W // 4,
              _lib.ptr(srcs[2]), H // 8, W // 8, n, c, _lib.stream())
    stats = torch.empty((n * H * W // 64, 3, c), device=DEV)
    _lib.call("diga_pyramid_sum_fwd3", _lib.ptr(b), H, W, _lib.ptr(bias), *[_lib.ptr(t) for t in srcs], _lib.ptr(stats), n, c, _lib.stream())
    # same taps in the same order; the compiler contracts the two kernels' expressions differently: last-bit differences
    assert_close(b, a, 1e-6, 2e-6, "tiled vs flat")
    # per chunk: sum (y - s), sum (y - s)^2, s  ->  total mean and (biased) variance
    sd, sd2, sh = stats[:, 0].double(), stats[:, 1].double(), stats[:, 2].double()
    m_chunk = sh + sd / 64.0
    mean = m_chunk.mean(0)
    ex2 = (sd2 / 64.0 + 2 * sh * (sd / 64.0) + sh * sh).mean(0)      # E[y^2] per chunk = E[(d + s)^2]
    var = ex2 - mean * mean
    y = b.double().reshape(-1, c)
    assert_close(mean, y.mean(0), 1e-5, 1e-5, "mean")
    assert_close(var, y.var(0, unbiased=False), 1e-4, 1e-5, "var")


def test_pyramid_sum_bwd_is_the_exact_adjoint():
    """<resize(s), g> == <s, resize^T(g)> to fp32 rounding at the benchmark's ratios (8, 4, 2) -- the property the gather relies on."""
    from diga_amd import _lib
    gen = synth.gen(7)
    n, c, H, W = 1, 8, 48, 64
    for r in (2, 4, 8):
        h, w = H // r, W // r
        s = torch.randn((n, h, w, c), generator=gen).to(DEV)
        g = torch.randn((n, H, W, c), generator=gen).to(DEV)
        up = torch.zeros((n, H, W, c), device=DEV)
        _lib.call("diga_pyramid_sum_fwd", _lib.ptr(up), H, W, None, _lib.ptr(s), h, w, None, 0, 0, None, 0, 0, n, c, _lib.stream())
        ds = torch.empty_like(s)
        _lib.call("diga_pyramid_sum_bwd", _lib.ptr(g), H, W, _lib.ptr(ds), h, w, n, c, _lib.stream())
        a, b = float((up.double() * g.double()).sum()), float((s.double() * ds.double()).sum())
        assert abs(a - b) < 1e-5 * max(abs(a), 1.0), (r, a, b)


def test_trainable_batchnorm_vs_torch():
    from diga_amd.model.norm import DigaTrainableBatchNorm2d
    gen = synth.gen(3)
    x = torch.randn((3, 40, 9, 11), generator=gen)
    probe = torch.randn((3, 40, 9, 11), generator=gen)
    ref = torch.nn.BatchNorm2d(40).double()
    mine = DigaTrainableBatchNorm2d(40)
    with torch.no_grad():
        ref.weight.copy_(1 + 0.2 * torch.randn(40, generator=gen))
        ref.bias.copy_(0.3 * torch.randn(40, generator=gen))
        mine.weight.copy_(ref.weight.float())
        mine.bias.copy_(ref.bias.float())
    mine = mine.to(DEV)
    for relu in (False, True):
        xr = x.double().requires_grad_()
        yr = ref(xr)
        yr = torch.relu(yr) if relu else yr
        ref.zero_grad()
        (yr * probe.double()).sum().backward()
        xm = x.to(DEV).requires_grad_()
        mine.zero_grad()
        ym = mine(xm, relu=relu)
        (ym * probe.to(DEV)).sum().backward()
        assert_close(ym, yr, 1e-5, 1e-5, "y")
        assert_close(xm.grad, xr.grad, 1e-4, 1e-5, "dx")
        assert_close(mine.weight.grad, ref.weight.grad, 1e-4, 1e-4, "dgamma")
        assert_close(mine.bias.grad, ref.bias.grad, 1e-4, 1e-4, "dbeta")
    assert_close(mine.running_mean, ref.running_mean, 1e-5, 1e-6, "running_mean")
    assert_close(mine.running_var, ref.running_var, 1e-5, 1e-6, "running_var")
    # eval mode (running statistics; BN-frozen fine-tuning of a head in .eval()): what nn.BatchNorm2d returns, no exception (round 5)
    mine.eval()
    ref.eval()
    xr = x.double().requires_grad_()
    ref.zero_grad()
    (torch.relu(ref(xr)) * probe.double()).sum().backward()
    xm = x.to(DEV).requires_grad_()
    mine.zero_grad()
    (mine(xm, relu=True) * probe.to(DEV)).sum().backward()
    assert_close(xm.grad, xr.grad, 1e-4, 1e-5, "eval dx")
    assert_close(mine.weight.grad, ref.weight.grad, 1e-4, 1e-4, "eval dgamma")
    assert_close(mine.bias.grad, ref.bias.grad, 1e-4, 1e-4, "eval dbeta")


def test_small_embed_ragged_vs_oracle(conv_math):
    """A second geometry against the oracle itself: embed_dim 64, 7 classes, maps of a 100x76 image, batch 3, train mode."""
    sd = oh.state_dict(CHANS, 7, 64)
    h = _head(embed=64, sd=sd, classes=7).train()
    gen = synth.gen(21)
    feats = [torch.randn((3, c, hh, ww), generator=gen) for c, (hh, ww) in zip(CHANS, ((25, 19), (13, 10), (7, 5), (4, 3)))]
    probe = torch.randn((3, 7, 25, 19), generator=gen)
    sdo = {k: (v.double().requires_grad_() if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    fo = [f.double().requires_grad_() for f in feats]
    lo, _, fused_o = oh.forward(sdo, fo, training=True)
    (lo * probe.double()).sum().backward()
    fd = [f.to(DEV).requires_grad_() for f in feats]
    lg, fused = h(fd)
    (lg * probe.to(DEV)).sum().backward()
    s = float(lo.abs().max())
    assert float((lg.detach().cpu().double() - lo.detach()).abs().max()) < TOL["logits"] * s
    assert float((fused.detach().cpu().double() - fused_o.detach()).abs().max()) < TOL["logits"] * float(fused_o.abs().max())
    for a, b in zip(fd, fo):
        assert float((a.grad.cpu().double() - b.grad).abs().max()) < TOL["dinput"] * float(b.grad.abs().max())
    for name, p in h.named_parameters():
        w = sdo[name].grad
        if name.endswith("proj.bias"):
            continue
        e = float((p.grad.detach().cpu().double() - w).abs().max() / w.abs().max())
        assert e < TOL["dparam"], (name, e)


def test_segformer_student_outputs_and_groups():
    from diga_amd.model.segformer import SegFormerStudent
    from oracle import mit as om
    m = SegFormerStudent("mit_b1")
    m.backbone.load_state_dict(om.state_dict(om.MIT_B1))
    m = m.to(DEV).train()
    x = torch.rand((2, 3, 128, 96), generator=synth.gen(1)).to(DEV) * 2 - 1
    c2, c4, logits, feat = m(x)
    assert tuple(c2.shape) == (2, 128, 16, 12) and tuple(c4.shape) == (2, 512, 4, 3)
    assert tuple(logits.shape) == (2, 19, 32, 24) and tuple(feat.shape) == (2, 768, 32, 24)
    logits.square().mean().backward()
    groups = m.optim_parameters(1e-3)
    head_ids = {id(p) for p in m.final.parameters()}
    assert {id(p) for p in groups[1]["params"]} == head_ids and groups[1]["lr"] == pytest.approx(1e-2)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
