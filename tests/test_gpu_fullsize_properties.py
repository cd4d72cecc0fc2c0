"""Parity at BASELINE.json's full sizes (config c2: B=8, 19 classes, 768x768, low-res 97x97; c4: 512x1024) through
size-independent properties -- the CPU oracle cannot finish these sizes in seconds, so the checks are identities
the operations must satisfy exactly or to rounding, plus spot checks of random pixels against the oracle."""
import random

import pytest
import torch

from conftest import assert_close
from oracle import losses as ol
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _labels(g, b, h, w):
    return synth.block_labels(g, b, h, w, block=32, ignore_frac=0.02)


def test_ce_fullsize_shift_invariance_and_gradient_rows():
    from diga_amd.util import loss as L
    g = synth.gen(1)
    B, C, H, W = 8, 19, 768, 768
    x = (2.0 * torch.randn((B, C, H, W), generator=g)).to(DEV)
    y = _labels(g, B, H, W).to(DEV)
    x.requires_grad_()
    l1 = L.cross_entropy2d(x, y)
    l1.backward()
    g1 = x.grad
    # softmax is invariant to a per-pixel shift of the logits
    shift = torch.randn((B, 1, H, W), generator=g).to(DEV)
    l2 = L.cross_entropy2d((x.detach() + shift), y)
    assert float(l2) == pytest.approx(float(l1), rel=2e-6)
    # gradient rows sum to zero over classes; ignored pixels carry exactly zero gradient
    assert float(g1.sum(dim=1).abs().max()) < 1e-9
    assert float(g1[(y == 255)[:, None].expand_as(g1)].abs().max()) == 0.0
    # spot check a crop against the oracle, with the full-size normalisation (all B*H*W pixels)
    xc, yc = x.detach()[:2, :, 100:164, 300:364].cpu(), y[:2, 100:164, 300:364].cpu()
    want = ol.ce_grad(xc, yc) * (2 * 64 * 64) / float(B * H * W)
    assert_close(g1[:2, :, 100:164, 300:364], want, 2e-5, 1e-12, "CE grad crop")


def test_distill_fullsize_properties():
    from diga_amd.util import loss as L
    g = synth.gen(2)
    B2, C, H, W = 16, 19, 768, 768
    t = (2.0 * torch.randn((B2, C, H, W), generator=g)).to(DEV)
    s = t.clone().requires_grad_()
    # student == teacher with the views swapped: loss = (1+scale) * mean cross-view soft CE; its gradient vanishes
    # exactly where both views agree, i.e. when the two halves of the batch are equal
    t_same = torch.cat([t[:8], t[:8]])
    s_same = t_same.clone().requires_grad_()
    l = L.distillation_loss(t_same, s_same)
    l.backward()
    assert float(s_same.grad.abs().max()) < 1e-9
    ent = -(torch.softmax(t[:8], 1) * torch.log_softmax(t[:8], 1)).sum(1).mean()
    assert float(l) == pytest.approx(1.5 * float(ent), rel=1e-5)
    # general case: per-pixel gradient rows sum to zero, the two views carry weights 0.5 and 1
    l2 = L.distillation_loss(t.flip(0), s)
    l2.backward()
    assert float(s.grad.sum(dim=1).abs().max()) < 1e-9
    r = float(s.grad[:8].abs().sum() / s.grad[8:].abs().sum())
    assert 0.45 < r < 0.55


@pytest.mark.parametrize("hw", [97, 192], ids=["deeplab_1_8", "segformer_1_4"])
def test_fused_loss_block_fullsize_vs_unfused_kernels(hw):
    """Low-res fused kernel (no full-res tensors) == explicit upsample + the two full-res loss kernels; logits at DeepLab's 1/8
    scale (one wave per 8 x 8-pixel cell) and at the SegFormer head's 1/4 scale (four 4 x 4-pixel cells per wave)."""
    from diga_amd import _lib
    from diga_amd.util import loss as L
    g = synth.gen(3)
    B, C, h, w, H, W = 8, 19, hw, hw, 768, 768
    stu = (2.0 * torch.randn((2 * B, C, h, w), generator=g)).to(DEV).requires_grad_()
    tea = (2.0 * torch.randn((2 * B, C, h, w), generator=g)).to(DEV)
    lab = _labels(g, B, H, W).to(DEV)
    total, ce, di = L.upsample_ce_distill(stu, tea, lab, 1.0, 0.5)
    total.backward()

    def up(x):
        y = torch.empty((x.shape[0], C, H, W), device=DEV)
        _lib.call("diga_upsample_bilinear_ac", _lib.ptr(x), _lib.ptr(y), x.shape[0] * C, h, w, H, W, _lib.stream())
        return y
    s_up, t_up = up(stu.detach()), up(tea)
    ce2 = L.cross_entropy2d(s_up[:B], lab)
    di2 = L.distillation_loss(t_up, s_up)
    assert float(ce) == pytest.approx(float(ce2), rel=1e-5)
    assert float(di) == pytest.approx(float(di2), rel=1e-5)
    # gradient: sum over the low-res gradient equals sum over the full-res one (bilinear weights sum to 1): 0 per pixel
    assert abs(float(stu.grad.sum())) < 1e-5
    # ... and per low-res pixel the class gradients sum to zero (every full-resolution softmax gradient does)
    assert float(stu.grad.sum(1).abs().max()) < 1e-5 * float(stu.grad.abs().max())
    # and a constant shift of the low-res logits changes nothing
    total2, _, _ = L.upsample_ce_distill(stu.detach() + 0.7, tea - 1.3, lab, 1.0, 0.5)
    assert float(total2) == pytest.approx(float(total), rel=1e-5)


def test_ema_sgd_fullsize_identities():
    from diga_amd.util import utils as U
    g = synth.gen(4)
    n = 65_063_568 // 64                       # same chunking code path, 1/64 of the parameter count per tensor
    shapes = [(n,), (2048, 512, 1, 1), (256, 2048, 3, 3)]
    stu = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    tea = [torch.nn.Parameter(p.detach().clone() + 1.0) for p in stu]

    class M(torch.nn.Module):
        def __init__(self, ps):
            super().__init__()
            self.ps = torch.nn.ParameterList(ps)
    ms, mt = M(stu), M(tea)
    with torch.no_grad():
        U.update_teacher_params(mt, ms, 0)      # alpha = 0: teacher := student, bit-exact
    assert all(torch.equal(a, b) for a, b in zip(mt.parameters(), ms.parameters()))
    with torch.no_grad():
        for p in mt.parameters():
            p.add_(1.0)
        before = [p.detach().clone() for p in mt.parameters()]
        U.update_teacher_params(mt, ms, 10 ** 9)   # alpha = 0.999: t <- 0.999 t + 0.001 s, elementwise
    for b, t_, s_ in zip(before, mt.parameters(), ms.parameters()):
        assert torch.equal(t_.detach(), 0.999 * b + (1 - 0.999) * s_.detach())
    # SGD: zero gradient, zero weight decay -> parameters unchanged; k duplicates of lr = one step of k*lr (momentum 0)
    q = [torch.nn.Parameter(p.detach().clone()) for p in stu]
    opt = U.DigaSGD([{"params": [q[0]] * 3 + [q[1]] * 2 + [q[2]]}], lr=0.1, momentum=0.0, weight_decay=0.0)
    for p in q:
        p.grad = torch.ones_like(p)
    opt.step()
    for p, p0, k in zip(q, stu, (3, 2, 1)):
        assert_close(p, p0.detach() - 0.1 * k, 1e-6, 1e-6, f"k={k} micro-steps")


def test_classmix_fullsize_identities():
    from diga_amd.util import utils as U
    g = synth.gen(5)
    B, H, W = 8, 768, 768
    labels = _labels(g, B, H, W).to(DEV)
    bg, fg = torch.randn((B, 3, H, W), generator=g).to(DEV), torch.randn((B, 3, H, W), generator=g).to(DEV)
    present = U.classmix_present(labels)
    assert all(p == sorted(set(p)) and 255 in p for p in present)
    all_sel = U.classmix_paste(bg, fg, labels, present)              # every class selected -> foreground
    assert torch.equal(all_sel, fg)
    none_sel = U.classmix_paste(bg, fg, labels, [[] for _ in range(B)])
    assert torch.equal(none_sel, bg)
    sels = U.classmix_select(present, random.Random(1))
    mixed, lab_out = U.classmix_paste(bg, fg, labels, sels, bg_labels=torch.full_like(labels, 7))
    inv = [[c for c in p if c not in s] for p, s in zip(present, sels)]
    other = U.classmix_paste(fg, bg, labels, inv)                    # complementary selection with swapped roles
    assert torch.equal(mixed, other)
    took = torch.stack([torch.isin(labels[b], torch.tensor(sels[b], device=DEV)) for b in range(B)])
    assert torch.equal(lab_out, torch.where(took, labels, torch.full_like(labels, 7)))


def test_conv_fullsize_linearity_and_adjoint():
    """Layer3 3x3 dilated conv at the c2 size: linearity in the input and <conv(x), y> == <x, conv^T(y)>
    (backward-data is the adjoint of forward; backward-weight the adjoint wrt the weights)."""
    from diga_amd.model.conv import DigaConv2d
    g = synth.gen(6)
    n, c, hw = 4, 256, 97
    m = DigaConv2d(c, c, 3, padding=2, dilation=2, bias=False).to(DEV)
    x1 = torch.randn((n, c, hw, hw), generator=g).to(DEV)
    x2 = torch.randn((n, c, hw, hw), generator=g).to(DEV)
    with torch.no_grad():
        y12, y1, y2 = m(x1 + x2), m(x1), m(x2)
    scale = float(y12.abs().max())
    # (three evaluations, each within the bound of the layer's kernel: 6x6 Winograd tiles for this geometry -- measured 2.8e-5)
    from conftest import WINO_TOL, winograd_tile
    assert float((y12 - y1 - y2).abs().max()) < max(2e-5, WINO_TOL[winograd_tile(n, c, hw, hw, c, 3, 1, 2, 2)][0]) * scale
    x = x1.clone().requires_grad_()
    yy = torch.randn(y1.shape, generator=g).to(DEV)
    out = m(x)
    (out * yy).sum().backward()
    lhs = float((out.detach().double() * yy.double()).sum())
    rhs_x = float((x1.double() * x.grad.double()).sum())
    rhs_w = float((m.weight.detach().double() * m.weight.grad.double()).sum())
    assert rhs_x == pytest.approx(lhs, rel=1e-4)
    assert rhs_w == pytest.approx(lhs, rel=1e-4)


@pytest.mark.parametrize("geom", [(256, 256, 3, 2), (1024, 256, 1, 1), (256, 1024, 1, 1)], ids=["l3.conv2", "l3.conv1", "l3.conv3"])
def test_conv_fullsize_split_bf16_vs_exact_fp32(geom):
    """The three layer-3 bottleneck convolutions at the true C2 geometry (16 images, 97x97): the split-bf16 kernels
    (256-row forward / backward-data tile on 1178+ blocks, 256-channel backward-weight tile with the pixel table and
    the one-round split-K plan) against the exact-fp32 MFMA kernels on the same inputs, the fused BatchNorm statistics
    against a direct reduction, and run-to-run bit reproducibility."""
    from diga_amd import _lib
    from diga_amd.model.conv import DigaConv2d
    cin, cout, k, dil = geom
    g = synth.gen(60 + cin // 64 + k)
    n, hw = 16, 97
    m = DigaConv2d(cin, cout, k, padding=dil * (k // 2), dilation=dil, bias=False).to(DEV)
    with torch.no_grad():
        m.weight.mul_(5.0)
    x0 = torch.randn((n, cin, hw, hw), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    yy = torch.randn((n, cout, hw, hw), generator=g).to(DEV).contiguous(memory_format=torch.channels_last)

    def run(mode):
        prev = _lib.get_conv_math()
        _lib.set_conv_math(mode)
        try:
            m.weight.grad = None
            m.emit_bn_stats = True
            m.train()
            x = x0.clone().requires_grad_()
            y = m(x)
            part = getattr(y, "_diga_bn_partials", None)          # (none in f32 when the layer takes the Winograd path)
            stats = part[0].clone() if part is not None else None
            (y * yy).sum().backward()
            return y.detach(), x.grad.detach(), m.weight.grad.detach().clone(), stats
        finally:
            _lib.set_conv_math(prev)

    y1, dx1, dw1, st1 = run(1)
    y0, dx0, dw0, _ = run(0)
    for a, b, what in ((y1, y0, "y"), (dx1, dx0, "dx"), (dw1, dw0, "dw")):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) < 1e-4 * scale, what
        assert float((a - b).norm() / b.norm()) < 3e-5, what
    # fused statistics: sum over the 128-row chunks of {sum d, shift} reproduces the column sums of y
    M = n * hw * hw
    nchunk = (M + 127) // 128
    st = st1[:nchunk * 3 * cout].view(nchunk, 3, cout).double()          # (the buffer has room for 64-row chunks)
    rows = torch.full((nchunk,), 128.0, dtype=torch.float64, device=DEV)
    rows[-1] = M - 128 * (nchunk - 1)
    col_sum = (st[:, 0] + st[:, 2] * rows[:, None]).sum(0)
    want = y1.permute(0, 2, 3, 1).reshape(M, cout).double().sum(0)
    assert float((col_sum - want).abs().max()) < 1e-6 * float(y1.abs().max()) * M
    y2, dx2, dw2, st2 = run(1)
    used = nchunk * 3 * cout                                              # (the rest of the buffer is room for 64-row chunks)
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and torch.equal(dw1, dw2) and torch.equal(st1[:used], st2[:used])


def test_batchnorm_fullsize_statistics():
    """Train-mode BN on a c2-sized layer1 tensor: the output has zero mean / unit variance per channel and the
    gradient is orthogonal to constants and to the normalised input (the two projections BN backward removes)."""
    from diga_amd.model.norm import DigaBatchNorm2d
    g = synth.gen(7)
    n, c, hw = 16, 64, 193
    bn = DigaBatchNorm2d(c)
    for p in bn.parameters():
        p.requires_grad = False
    bn = bn.to(DEV).train()
    x = (3.0 * torch.randn((n, c, hw, hw), generator=g) + 5.0).to(DEV).contiguous(memory_format=torch.channels_last)
    x.requires_grad_()
    y = bn(x)
    assert float(y.mean(dim=(0, 2, 3)).abs().max()) < 1e-4
    assert float((y.var(dim=(0, 2, 3), unbiased=False) - 1).abs().max()) < 1e-3
    probe = torch.randn(y.shape, generator=g).to(DEV)
    (y * probe).sum().backward()
    gx = x.grad
    assert float(gx.sum(dim=(0, 2, 3)).abs().max()) < 5e-2 * float(gx.abs().sum(dim=(0, 2, 3)).max()) * 1e-3
    assert float((gx * y.detach()).sum(dim=(0, 2, 3)).abs().max()) < 5e-2 * float(gx.abs().sum(dim=(0, 2, 3)).max()) * 1e-3


def test_centroid_pipeline_fullsize_c4():
    """c4-sized pseudo-labeler: weights are a softmax (sum to 1), features equal to a centroid get that centroid's
    label, the consensus output only removes labels, class sums add up to the feature total of member pixels."""
    from diga_amd.calc_centroids import Class_Features
    g = synth.gen(8)
    N, D, h, w, H, W = 8, 256, 65, 129, 512, 1024
    cents = torch.randn((19, D), generator=g)
    cls_map = torch.randint(0, 19, (N, h, w), generator=g)
    feat = (cents[cls_map].permute(0, 3, 1, 2).contiguous() + 0.05 * torch.randn((N, D, h, w), generator=g)).to(DEV)
    cf = Class_Features(numbers=19)
    cf.objective_vectors = cents.to(DEV)
    wts = cf.get_centroid_weight(feat)
    assert float((wts.sum(1) - 1).abs().max()) < 1e-5
    assert torch.equal(wts.argmax(1).cpu(), cls_map)
    pseudo_prob = _labels(g, N, H, W).to(DEV)
    out, fp = cf.consensus_pseudo_labels(feat, pseudo_prob, return_feat_pseudo=True)
    changed = out != pseudo_prob
    assert bool((out[changed] == 255).all())                          # the filter only ever removes labels
    assert bool((pseudo_prob[changed] != fp[changed]).all())          # ... and only where the two labels disagree
    assert bool((out[(pseudo_prob == fp)] == pseudo_prob[(pseudo_prob == fp)]).all())
    logits = torch.nn.functional.one_hot(cls_map, 19).permute(0, 3, 1, 2).float().to(DEV) * 5
    sums, counts, hw = cf._class_sums(feat, logits)
    assert int(counts.sum()) == N * h * w
    assert_close(sums.sum(dim=1), feat.sum(dim=(2, 3)), 1e-4, 1e-2, "class sums partition the feature total")
