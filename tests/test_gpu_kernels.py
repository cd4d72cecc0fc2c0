"""GPU parity tests: the HIP path (through the C ABI, via the reference-shaped host modules) against
the committed golden captures of the reference and against the CPU oracle on seeded inputs.
Run on the MI355X box with  python -m pytest tests -m gpu."""
import random

import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import centroids as oc
from oracle import classmix as ocm
from oracle import losses as ol
from oracle import metrics as om
from oracle import optim as oo
from oracle import synth

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def mods():
    from diga_amd import calc_centroids, _lib
    from diga_amd.util import loss, metrics, utils
    return dict(loss=loss, utils=utils, metrics=metrics, cc=calc_centroids, lib=_lib)


# ----------------------------------------------------------------------------- cross_entropy2d
def _ce_case(mods, x_cpu, y_cpu, want_loss, want_grad, rtol=2e-5):
    x = x_cpu.to(DEV).requires_grad_()
    loss = mods["loss"].cross_entropy2d(x, y_cpu.to(DEV))
    loss.backward()
    assert_close(loss, want_loss, 1e-5, 1e-7, "ce loss")
    assert_close(x.grad, want_grad, rtol, 1e-9, "ce grad")


def test_ce_golden(mods, golden):
    g = golden("ce")
    _ce_case(mods, g.t("kat_x"), g.t("kat_y"), g.t("kat_loss"), g.t("kat_grad"))      # KAT-1, W=4: vector path
    _ce_case(mods, g.t("x"), g.t("y"), g.t("loss"), g.t("grad"))                      # 33x33: scalar path
    _ce_case(mods, g.t("allign_x"), g.t("allign_y"), g.t("allign_loss"), g.t("allign_grad"))


@pytest.mark.parametrize("shape,C", [((3, 64, 96), 19), ((2, 37, 53), 19), ((2, 32, 32), 16), ((1, 16, 24), 7)])
def test_ce_vs_oracle(mods, shape, C):
    g = synth.gen(sum(shape) + C)
    n, h, w = shape
    x = 4.0 * torch.randn((n, C, h, w), generator=g)
    y = torch.randint(0, C, shape, generator=g)
    y[torch.rand(shape, generator=g) < 0.1] = 255
    xo = x.clone().requires_grad_()
    lo = ol.cross_entropy2d(xo, y)
    lo.backward()
    _ce_case(mods, x, y, lo.detach(), xo.grad)


def test_ce_upstream_scale_and_loss_only(mods):
    g = synth.gen(5)
    x = torch.randn((2, 19, 16, 16), generator=g)
    y = torch.randint(0, 19, (2, 16, 16), generator=g)
    xd = x.to(DEV).requires_grad_()
    (0.37 * mods["loss"].cross_entropy2d(xd, y.to(DEV))).backward()
    assert_close(xd.grad, 0.37 * ol.ce_grad(x, y), 2e-5, 1e-9, "scaled grad")
    with torch.no_grad():
        l = mods["loss"].cross_entropy2d(x.to(DEV), y.to(DEV))
    assert_close(l, ol.cross_entropy2d(x, y), 1e-5, 1e-7)
    ls = mods["loss"].cross_entropy2d(x.to(DEV), y.to(DEV), size_average=False)
    assert_close(ls, ol.cross_entropy2d(x, y, size_average=False), 1e-5, 1e-4)


# ----------------------------------------------------------------------------- OhemCrossEntropy ("next" row 4)
def _ohem_check(loss, grad, want_loss, want_grad, what):
    assert_close(loss, want_loss, 2e-5, 1e-7, what + " loss")
    # a pixel whose probability is within an ulp of the threshold may fall on the other side: compare in norm too
    scale = float(want_grad.abs().max())
    err = (grad.cpu() - want_grad).abs()
    assert float((err > 2e-5 * scale + 1e-9).float().mean()) < 1e-3, what + " grad: too many pixels differ"
    assert float(err.norm() / max(float(want_grad.norm()), 1e-30)) < 1e-3, what + " grad norm"


@pytest.mark.parametrize("tag,kw", [("a", dict(min_kept=50)), ("b", dict(min_kept=600)), ("c", dict(min_kept=100000)),
                                    ("d", dict(min_kept=300, thres=0.5))])
def test_ohem_golden(mods, golden, tag, kw):
    g = golden("ohem")
    src = "b" if tag == "c" else tag
    x = g.t(src + "_x").to(DEV).requires_grad_()
    loss = mods["loss"].OhemCrossEntropy(**kw)(x, g.t(src + "_y").to(DEV))
    loss.backward()
    _ohem_check(loss, x.grad, g.t(tag + "_loss"), g.t(tag + "_grad"), f"ohem {tag}")


@pytest.mark.parametrize("shape,C,min_kept,conf", [((2, 96, 128), 19, 5000, 5.0), ((3, 37, 53), 19, 100000, 0.0),
                                                  ((1, 64, 64), 16, 100, 8.0), ((2, 128, 256), 19, 20000, 3.0)])
def test_ohem_vs_oracle(mods, shape, C, min_kept, conf):
    g = synth.gen(sum(shape) + C + min_kept)
    n, h, w = shape
    y = torch.randint(0, C, shape, generator=g)
    x = torch.randn((n, C, h, w), generator=g)
    x = x + conf * torch.nn.functional.one_hot(y, C).permute(0, 3, 1, 2).float() * (torch.rand((n, 1, h, w), generator=g) < 0.8)
    y[torch.rand(shape, generator=g) < 0.1] = 255
    xo = x.clone().requires_grad_()
    lo, kept, thr = ol.ohem_cross_entropy(xo, y, min_kept=min_kept)
    lo.backward()
    xd = x.to(DEV).requires_grad_()
    loss = mods["loss"].OhemCrossEntropy(min_kept=min_kept)(xd, y.to(DEV))
    (2.0 * loss).backward()
    _ohem_check(loss, xd.grad / 2.0, lo.detach(), xo.grad, f"ohem {shape}")
    # bit-reproducible: integer-atomic histograms and fixed-order sums
    xe = x.to(DEV).requires_grad_()
    loss2 = mods["loss"].OhemCrossEntropy(min_kept=min_kept)(xe, y.to(DEV))
    (2.0 * loss2).backward()
    assert torch.equal(loss, loss2) and torch.equal(xd.grad, xe.grad)


def test_ohem_all_ignored_is_nan_with_zero_grad(mods):
    x = torch.randn((1, 19, 8, 8)).to(DEV).requires_grad_()
    y = torch.full((1, 8, 8), 255, dtype=torch.int64, device=DEV)
    loss = mods["loss"].OhemCrossEntropy()(x, y)
    loss.backward()
    assert torch.isnan(loss) and float(x.grad.abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        mods["loss"].OhemCrossEntropy(weight=torch.ones(19))


# ----------------------------------------------------------------------------- distillation_loss
def test_distill_golden(mods, golden):
    g = golden("distill")
    for tk, sk, lk, gk, scale in (("kat_t", "kat_s", "kat_loss", "kat_grad", 0.5), ("t", "s", "loss", "grad", 0.5),
                                  ("t", "s", "loss_q", "grad_q", 0.25)):
        s = g.t(sk).to(DEV).requires_grad_()
        loss = mods["loss"].distillation_loss(g.t(tk).to(DEV), s, scale)
        loss.backward()
        assert_close(loss, g.t(lk), 1e-5, 1e-7, lk)
        assert_close(s.grad, g.t(gk), 2e-5, 2e-6 * float(g.t(gk).abs().max()), gk)


@pytest.mark.parametrize("shape,C", [((4, 48, 64), 19), ((2, 33, 35), 19), ((6, 16, 16), 16), ((2, 8, 8), 5)])
def test_distill_vs_oracle(mods, shape, C):
    g = synth.gen(11 + C)
    b2, h, w = shape
    t = 3.0 * torch.randn((b2, C, h, w), generator=g)
    s = 3.0 * torch.randn((b2, C, h, w), generator=g)
    so = s.clone().requires_grad_()
    lo = ol.distillation_loss(t, so)
    lo.backward()
    sd = s.to(DEV).requires_grad_()
    ld = mods["loss"].distillation_loss(t.to(DEV), sd)
    (2.0 * ld).backward()
    assert_close(ld, lo.detach(), 1e-5, 1e-7, "loss")
    assert_close(sd.grad, 2.0 * so.grad, 2e-5, 4e-6 * float(so.grad.abs().max()), "grad")


# ----------------------------------------------------------------------------- fused upsample + losses
def test_fused_loss_block_golden(mods, golden):
    g = golden("upsample")
    stu = g.t("stu").to(DEV).requires_grad_()
    total, ce, di = mods["loss"].upsample_ce_distill(stu, g.t("tea").to(DEV), g.t("lab").to(DEV))
    total.backward()
    assert_close(ce, g.t("ce"), 1e-5, 1e-7, "ce")
    assert_close(di, g.t("distil"), 1e-5, 1e-7, "distil")
    assert_close(total, g.t("total"), 1e-5, 1e-7, "total")
    assert_close(stu.grad, g.t("grad_stu"), 2e-4, 3e-8, "grad wrt low-res student")


@pytest.mark.parametrize("B,hw,HW", [(2, (17, 17), (128, 128)), (1, (9, 13), (65, 97)), (3, (5, 7), (33, 50)),
                                     (2, (33, 33), (256, 256)), (1, (6, 6), (6, 6)), (1, (8, 8), (5, 5)),
                                     # small cells: 4, 2 and 8 neighbouring cells share a wave (logits at 1/4 scale: the SegFormer head)
                                     (2, (32, 32), (128, 128)), (1, (24, 40), (96, 120)), (2, (33, 21), (128, 128)), (1, (48, 50), (96, 100))])
def test_fused_loss_block_vs_oracle(mods, B, hw, HW):
    g = synth.gen(B * 100 + hw[0] + HW[1])
    stu = 2.0 * torch.randn((2 * B, 19, *hw), generator=g)
    tea = 2.0 * torch.randn((2 * B, 19, *hw), generator=g)
    lab = torch.randint(0, 19, (B, *HW), generator=g)
    lab[torch.rand((B, *HW), generator=g) < 0.15] = 255
    so = stu.clone().requires_grad_()
    total, ce, di = ol.warmup_losses_lowres(so, tea, lab, 1.0, 0.5)
    total.backward()
    sd = stu.to(DEV).requires_grad_()
    t2, c2, d2 = mods["loss"].upsample_ce_distill(sd, tea.to(DEV), lab.to(DEV), 1.0, 0.5)
    t2.backward()
    assert_close(c2, ce.detach(), 2e-5, 1e-7, "ce")
    assert_close(d2, di.detach(), 2e-5, 1e-7, "distil")
    assert_close(sd.grad, so.grad, 3e-4, 5e-8, "grad")
    # run-to-run determinism (no float atomics)
    sd2 = stu.to(DEV).requires_grad_()
    t3, _, _ = mods["loss"].upsample_ce_distill(sd2, tea.to(DEV), lab.to(DEV), 1.0, 0.5)
    t3.backward()
    assert torch.equal(sd.grad, sd2.grad) and torch.equal(t2, t3)


def test_aux_head_loss_block_vs_oracle(mods):
    """Two-headed (HRNet-OCR) warm-up loss of the semi-supervised tree, train_DiGA_semiseg_warm_up.py:259-263,282:
    CE + 0.1 CE_aux and distill + 0.1 distill_aux on full-resolution upsampled logits, against the oracle composition."""
    g = synth.gen(4100)
    B, hw, HW = 2, (17, 33), (129, 257)
    maps = [2.0 * torch.randn((2 * B, 19, *hw), generator=g) for _ in range(4)]     # stu main, stu aux, tea main, tea aux
    lab = torch.randint(0, 19, (B, *HW), generator=g)
    lab[torch.rand((B, *HW), generator=g) < 0.1] = 255
    sm, sa = maps[0].clone().requires_grad_(), maps[1].clone().requires_grad_()
    up = lambda t: ol.upsample_bilinear_ac(t, HW)  # noqa: E731
    semseg = ol.cross_entropy2d(up(sm)[:B], lab) + 0.1 * ol.cross_entropy2d(up(sa)[:B], lab)
    distil = ol.distillation_loss(up(maps[2]), up(sm)) + 0.1 * ol.distillation_loss(up(maps[3]), up(sa))
    (1.0 * semseg + 0.5 * distil).backward()
    dm, da = maps[0].to(DEV).requires_grad_(), maps[1].to(DEV).requires_grad_()
    total, ce, di = mods["loss"].upsample_ce_distill_aux(dm, da, maps[2].to(DEV), maps[3].to(DEV), lab.to(DEV), 1.0, 0.5, 0.1)
    total.backward()
    assert_close(ce, semseg.detach(), 2e-5, 1e-7, "loss_semseg")
    assert_close(di, distil.detach(), 2e-5, 1e-7, "loss_distil")
    assert_close(total, (semseg + 0.5 * distil).detach(), 2e-5, 1e-7, "total")
    assert_close(dm.grad, sm.grad, 3e-4, 5e-8, "grad main head")
    assert_close(da.grad, sa.grad, 3e-4, 5e-8, "grad aux head")


def test_fused_ce_only_vs_oracle(mods):
    g = synth.gen(77)
    lr = 2.0 * torch.randn((3, 19, 9, 17), generator=g)
    lab = torch.randint(0, 19, (3, 65, 129), generator=g)
    lab[:, ::7] = 255
    lo = lr.clone().requires_grad_()
    want = ol.cross_entropy2d(ol.upsample_bilinear_ac(lo, (65, 129)), lab)
    want.backward()
    ld = lr.to(DEV).requires_grad_()
    got = mods["loss"].upsample_ce(ld, lab.to(DEV))
    got.backward()
    assert_close(got, want.detach(), 2e-5, 1e-7)
    assert_close(ld.grad, lo.grad, 3e-4, 5e-8)


def test_upsample_bilinear(mods, golden):
    g = golden("upsample")
    lib = mods["lib"]
    x = g.t("x").to(DEV)
    y = torch.empty((2, 19, 65, 97), device=DEV)
    lib.call("diga_upsample_bilinear_ac", lib.ptr(x), lib.ptr(y), 2 * 19, 9, 13, 65, 97, lib.stream())
    assert_close(y, g.t("y"), 1e-5, 2e-6, "upsample")


# ----------------------------------------------------------------------------- EMA
class _Net(torch.nn.Module):
    def __init__(self, tensors, buf):
        super().__init__()
        for i, t in enumerate(tensors):
            setattr(self, f"p{i}", torch.nn.Parameter(t.clone()))
        self.register_buffer("buf", buf.clone())


def test_ema_golden_bit_exact(mods, golden):
    g = golden("ema")
    U = mods["utils"]
    stu = _Net([g.t("s_a"), g.t("s_b"), g.t("s_c")], torch.zeros(4)).to(DEV)
    tea = _Net([g.t("t_a"), g.t("t_b"), g.t("t_c")], g.t("t_buf")).to(DEV)
    for k, it in enumerate(g["its"].tolist()):
        with torch.no_grad():
            for p in stu.parameters():
                p.add_(0.01 * (k + 1))
            U.update_teacher_params(tea, stu, it)
        for name, p in zip("abc", tea.parameters()):
            assert torch.equal(p.detach().cpu(), g.t(f"t_{name}_{k}")), f"EMA not bit-exact at it={it} ({name})"
    assert torch.equal(tea.buf.cpu(), g.t("t_buf"))                     # buffers untouched
    fresh = _Net([torch.zeros(7, 5), torch.zeros(1031), torch.zeros(3, 4, 3, 3)], torch.ones(4)).to(DEV)
    U.create_teacher_params(fresh, stu)
    assert all(torch.equal(a, b) for a, b in zip(fresh.parameters(), stu.parameters()))
    assert torch.equal(fresh.buf.cpu(), torch.ones(4))


def test_ema_many_tensors_and_flat(mods):
    g = synth.gen(21)
    U, lib = mods["utils"], mods["lib"]
    shapes = [(1,), (3,), (16385,), (64, 3, 7, 7), (40000,), (5, 5), (2,)]
    s = [torch.randn(sh, generator=g) for sh in shapes]
    t = [torch.randn(sh, generator=g) for sh in shapes]
    stu, tea = _Net(s, torch.zeros(1)).to(DEV), _Net(t, torch.zeros(1)).to(DEV)
    with torch.no_grad():
        U.update_teacher_params(tea, stu, 57)
    tt = [x.clone() for x in t]
    oo.ema_update(tt, s, 57)
    for a, b in zip(tea.parameters(), tt):
        assert torch.equal(a.detach().cpu(), b)
    n = 1_000_003
    fs, ft = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ds, dt = fs.to(DEV), ft.to(DEV)
    a = oo.ema_alpha(3)
    lib.call("diga_ema_update_flat", lib.ptr(dt), lib.ptr(ds), n, float(a), float(1 - a), lib.stream())
    want = [ft.clone()]
    oo.ema_update(want, [fs], 3)
    assert torch.equal(dt.cpu(), want[0])


# ----------------------------------------------------------------------------- SGD
def test_sgd_duplicates_golden(mods, golden):
    g = golden("sgd")
    U = mods["utils"]
    mults, groups = g["mults"].tolist(), g["groups"].tolist()
    params = [torch.nn.Parameter(g.t(f"p0_{i}").to(DEV)) for i in range(4)]
    g1 = [p for p, m, gr in zip(params, mults, groups) if gr == 0 for _ in range(m)]
    g10 = [p for p, m, gr in zip(params, mults, groups) if gr == 1 for _ in range(m)]
    opt = U.DigaSGD([{"params": g1, "lr": 2.5e-4}, {"params": g10, "lr": 2.5e-3}], lr=2.5e-4, momentum=0.9,
                    weight_decay=5e-4)
    for step in range(3):
        U.adjust_learning_rate([opt], base_lr=2.5e-4, i_iter=step, max_iter=100, power=0.9)
        assert opt.param_groups[0]["lr"] == pytest.approx(float(g[f"lr_{step}"][0]), rel=1e-12)
        for i, p in enumerate(params):
            p.grad = g.t(f"g{step}_{i}").to(DEV)
        opt.step()
        bufs = opt.momentum_buffers()
        for i, p in enumerate(params):
            assert_close(p, g.t(f"p{step + 1}_{i}"), 1e-6, 1e-7, f"param {i} step {step}")
            assert_close(bufs[id(p)], g.t(f"buf{step + 1}_{i}"), 1e-6, 1e-7, f"buf {i} step {step}")
    q = torch.nn.Parameter(torch.ones(1, device=DEV))
    o2 = U.DigaSGD([{"params": [q, q, q]}], lr=0.1, momentum=0.9, weight_decay=0.01)
    for want in g["kat_scalar"].tolist():
        q.grad = torch.full((1,), 2.0, device=DEV)
        o2.step()
        assert float(q) == pytest.approx(want, rel=1e-6)


def test_sgd_large_vs_oracle(mods):
    g = synth.gen(31)
    U = mods["utils"]
    shapes, mults = [(100003,), (64, 64, 3, 3), (7,)], [3, 4, 1]
    p0 = [torch.randn(s, generator=g) for s in shapes]
    params = [torch.nn.Parameter(p.to(DEV)) for p in p0]
    opt = U.DigaSGD([{"params": [p for p, m in zip(params, mults) for _ in range(m)]}], lr=1e-2, momentum=0.9,
                    weight_decay=5e-4, grad_scale=0.5)
    ref = [p.clone() for p in p0]
    bufs = [torch.zeros_like(p) for p in p0]
    for step in range(2):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for p, gr in zip(params, grads):
            p.grad = gr.to(DEV)
        opt.step()
        oo.sgd_step_dup(ref, [0.5 * gr for gr in grads], bufs, mults, [1e-2] * 3, first_step=(step == 0))
        for p, r in zip(params, ref):
            assert_close(p, r, 2e-6, 1e-7, f"step {step}")


def test_nonfinite_flag_and_sgd_skip(mods):
    """diga_nonfinite_flag_f32 (inf / NaN anywhere in a tensor -> flag[0] = 1, flag[1] += 1, including the ragged tail) and the
    skip_flag argument of diga_sgd_momentum_multi: a set flag leaves parameters and momentum bit-identical."""
    from diga_amd import _lib
    U = mods["utils"]
    flag = torch.zeros(2, dtype=torch.int32, device=DEV)
    x = torch.randn(100003, generator=synth.gen(5)).to(DEV)
    _lib.call("diga_nonfinite_flag_f32", _lib.ptr(x), x.numel(), _lib.ptr(flag), _lib.stream())
    assert flag.cpu().tolist() == [0, 0]
    for pos, val in ((100002, float("inf")), (17, float("nan")), (51234, float("-inf"))):
        y = x.clone()
        y[pos] = val
        flag[0] = 0
        _lib.call("diga_nonfinite_flag_f32", _lib.ptr(y), y.numel(), _lib.ptr(flag), _lib.stream())
        assert int(flag[0]) == 1, (pos, val)
    assert int(flag[1]) == 3
    _lib.call("diga_nonfinite_flag_f32", _lib.ptr(y), y.numel(), _lib.ptr(flag), _lib.stream())      # already set: counted once
    assert flag.cpu().tolist() == [1, 3]
    p = torch.nn.Parameter(torch.randn(1000, generator=synth.gen(6)).to(DEV))
    opt = U.DigaSGD([{"params": [p, p]}], lr=0.1, momentum=0.9, weight_decay=0.0)
    p.grad = torch.ones_like(p)
    opt.step()
    p1, b1 = p.detach().clone(), opt.momentum_buffers()[id(p)].clone()
    p.grad = torch.full_like(p, float("nan"))
    opt.step(found_inf=flag)                                         # flag[0] == 1: nothing moves
    assert torch.equal(p.detach(), p1) and torch.equal(opt.momentum_buffers()[id(p)], b1)
    flag.zero_()
    p.grad = torch.ones_like(p)
    opt.step(found_inf=flag)
    assert not torch.equal(p.detach(), p1)


# ----------------------------------------------------------------------------- ClassMix
def test_classmix_golden(mods, golden):
    g = golden("classmix")
    U = mods["utils"]
    labels = g.t("labels").to(DEV)
    present = U.classmix_present(labels)
    for b in range(2):
        assert present[b] == ocm.classes_present(g.t("labels")[b])
    random.seed(int(g["seed"][0]))
    mixed, mixed_lab, sels = U.classmix(g.t("bg").to(DEV), g.t("fg").to(DEV), labels, random,
                                        bg_labels=g.t("bg_lab").to(DEV))
    for b, sel in enumerate(sels):
        assert sel == [v for v in g["sel"][b].tolist() if v >= 0]
    assert torch.equal(mixed.cpu(), g.t("mixed"))                    # bit-exact
    assert torch.equal(mixed_lab.cpu(), g.t("mixed_lab"))


@pytest.mark.parametrize("B,H,W", [(3, 64, 64), (2, 37, 51), (1, 1, 5)])
def test_classmix_vs_oracle(mods, B, H, W):
    g = synth.gen(B + H + W)
    U = mods["utils"]
    labels = synth.block_labels(g, B, H, W, block=8, ignore_frac=0.05)
    bg, fg = torch.randn((B, 3, H, W), generator=g), torch.randn((B, 3, H, W), generator=g)
    random.seed(9)
    want, mask, sels = ocm.classmix(bg, fg, labels, random)
    random.seed(9)
    got, sels2 = U.classmix(bg.to(DEV), fg.to(DEV), labels.to(DEV), random)
    assert sels == sels2
    assert torch.equal(got.cpu(), want)
    hist = torch.zeros((B, 256), dtype=torch.int32, device=DEV)
    lib = mods["lib"]
    lab = labels.to(DEV)
    lib.call("diga_label_hist256", lib.ptr(lab), lib.ptr(hist), B, H * W, lib.stream())
    want_hist = torch.stack([torch.bincount(labels[b].reshape(-1), minlength=256) for b in range(B)])
    assert torch.equal(hist.cpu().long(), want_hist)


# ----------------------------------------------------------------------------- centroids
def test_centroid_weights_golden(mods, golden):
    g = golden("centroid")
    cf = mods["cc"].Class_Features(numbers=19)
    cf.objective_vectors = g.t("kat_c").to(DEV)
    w = cf.get_centroid_weight(g.t("kat_f").to(DEV))
    assert_close(w, g.t("kat_w"), 3e-4, 1e-12, "KAT-3 weights")
    cf.objective_vectors = g.t("cents").to(DEV)
    feat = g.t("feat").to(DEV)
    assert_close(cf.get_centroid_weight(feat), g.t("w"), 3e-4, 1e-9, "weights")
    assert_close(cf.get_centroid_distance(feat), g.t("neg_dist"), 1e-5, 1e-5, "neg distance")
    assert_close(cf.feat_centroid_distance(feat), -g.t("neg_dist"), 1e-5, 1e-5, "distance")
    pseudo, fp = cf.consensus_pseudo_labels(feat, g.t("pseudo_prob").to(DEV), return_feat_pseudo=True)
    safe = g.t("margin") > 1e-5
    assert bool((fp.cpu() == g.t("feat_pseudo"))[safe].all())
    assert bool((pseudo.cpu() == g.t("pseudo"))[safe].all())


def test_class_means_golden(mods, golden):
    g = golden("meanvec")
    feat, out, cents = g.t("feat").to(DEV), g.t("out").to(DEV), g.t("cents")
    for tag, lab in (("nolab", None), ("lab", g.t("lab_lr").to(DEV))):
        cf = mods["cc"].Class_Features(numbers=19)
        cf.objective_vectors = cents.clone().to(DEV)
        vecs, ids = cf.calculate_mean_vector(feat, out, lab)
        assert ids == g[f"{tag}_ids"].tolist()
        assert_close(torch.stack([v.reshape(-1) for v in vecs]), g.t(f"{tag}_vecs"), 2e-5, 2e-6, f"{tag} vectors")
        for t in range(len(ids)):                                    # reference-style sequential updates
            cf.update_objective_SingleVector(ids[t], vecs[t].detach(), start_mean=False)
        assert_close(cf.objective_vectors, g.t(f"{tag}_cents"), 1e-6, 1e-7, f"{tag} centroids (single)")
        assert torch.equal(cf.objective_vectors_num.cpu(), g.t(f"{tag}_nums"))
        cf2 = mods["cc"].Class_Features(numbers=19)                  # device-resident fast path
        cf2.objective_vectors = cents.clone().to(DEV)
        cf2.update_from_batch(feat, out, labels_lr=lab)
        assert_close(cf2.objective_vectors, g.t(f"{tag}_cents"), 1e-6, 1e-7, f"{tag} centroids (batch)")
        assert torch.equal(cf2.objective_vectors_num.cpu(), g.t(f"{tag}_nums"))
    cf = mods["cc"].Class_Features(numbers=19)                       # 'mean' mode of the offline pass
    vecs, ids = cf.calculate_mean_vector(feat, out)
    for _ in range(2):
        for t in range(len(ids)):
            cf.update_objective_SingleVector(ids[t], vecs[t].detach().cpu().numpy(), "mean")
    assert_close(cf.objective_vectors, g.t("mean_cents"), 1e-5, 1e-6, "mean-mode centroids")
    assert torch.equal(cf.objective_vectors_num.cpu(), g.t("mean_nums"))


def test_class_means_fullres_labels_vs_oracle(mods):
    g = synth.gen(41)
    N, D, h, w, H, W = 3, 256, 33, 65, 256, 512
    feat = torch.randn((N, D, h, w), generator=g)
    out = torch.randn((N, 19, h, w), generator=g)
    blocks = torch.randint(0, 19, (N, 5, 9), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :h, :w]
    out.scatter_add_(1, blocks[:, None], torch.full((N, 1, h, w), 6.0))
    lab = synth.block_labels(g, N, H, W, block=64, ignore_frac=0.05)
    lab_lr = oc.nearest_downsample_labels(lab, (h, w))
    take = torch.rand((N, h, w), generator=g) < 0.5
    lab_full = lab.clone()
    # make half of the low-res sample points agree with the prediction
    ys = (torch.arange(h, dtype=torch.float32) * (H / h)).floor().long().clamp_(max=H - 1)
    xs = (torch.arange(w, dtype=torch.float32) * (W / w)).floor().long().clamp_(max=W - 1)
    sub = lab_full[:, ys][:, :, xs]
    sub[take] = out.argmax(1)[take]
    lab_full[:, ys[:, None], xs[None, :]] = sub
    lab_lr = oc.nearest_downsample_labels(lab_full, (h, w))
    vecs, ids, owners = oc.class_mean_vectors(feat, out, lab_lr)
    cents = torch.randn((19, D), generator=g)
    c, n = cents.clone(), torch.zeros(19)
    oc.centroid_ema_apply(c, n, vecs, ids)
    cf = mods["cc"].Class_Features(numbers=19)
    cf.objective_vectors = cents.clone().to(DEV)
    sums, counts = cf.update_from_batch(feat.to(DEV), out.to(DEV), labels_full=lab_full.to(DEV))
    assert len(ids) > 10
    assert_close(cf.objective_vectors, c, 1e-6, 1e-7, "centroids after EMA")
    assert torch.equal(cf.objective_vectors_num.cpu(), n)
    cnt = counts.cpu()
    for v, t, o in zip(vecs, ids, owners):
        assert_close(sums[o, t].cpu() / float(cnt[o, t]), v, 3e-5, 3e-6, f"mean vector n={o} t={t}")


def test_consensus_vs_oracle_large(mods):
    g = synth.gen(43)
    cents = torch.randn((19, 256), generator=g)
    cls_map = torch.randint(0, 19, (2, 33, 65), generator=g)
    feat = cents[cls_map].permute(0, 3, 1, 2).contiguous() * 0.6 + 0.7 * torch.randn((2, 256, 33, 65), generator=g)
    pseudo_prob = synth.block_labels(g, 2, 256, 512, block=16, ignore_frac=0.05)
    w = oc.centroid_weight(feat, cents)
    want, want_fp = oc.consensus_filter(w, pseudo_prob)
    up = ol.upsample_bilinear_ac(w, (256, 512))
    top2 = up.topk(2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-5
    cf = mods["cc"].Class_Features(numbers=19)
    cf.objective_vectors = cents.to(DEV)
    got, fp = cf.consensus_pseudo_labels(feat.to(DEV), pseudo_prob.to(DEV), return_feat_pseudo=True)
    assert bool((fp.cpu() == want_fp)[safe].all()) and bool((got.cpu() == want)[safe].all())
    assert float(safe.float().mean()) > 0.999


@pytest.mark.parametrize("N,D,hw,HW", [(1, 256, (17, 23), (130, 182)),      # ragged block edges of the pair kernel (182 = 128 + 54, 130 = 16 * 8 + 2)
                                       (2, 256, (17, 23), (129, 181)),      # odd width: the one-pixel-per-thread kernel
                                       (1, 100, (9, 10), (36, 40)),         # D = 100: a wave's quarter is 25 channels (one 16-deep round + a tail of 9), x4 upsampling
                                       (2, 64, (33, 65), (66, 130)),        # x2 upsampling: outside the pair kernel's tile (falls back)
                                       (1, 256, (5, 7), (5, 7))])           # identity geometry
def test_centroid_weights_and_consensus_edge_shapes_vs_oracle(mods, N, D, hw, HW):
    """Round 6 rebuilt the centroid-weights kernel (packed class pairs, 16-deep load rounds + a channel tail) and the consensus kernel
    (128 x 8 blocks of pixel pairs, class-fastest LDS tile, with the round-5 kernel as the fall-back for odd widths / small upsampling
    factors): geometries that hit the ragged edges, the tail loop and both fall-backs, against the oracle away from argmax near-ties."""
    g = synth.gen(N * 1000 + D + hw[0] + HW[1])
    cents = torch.randn((19, D), generator=g)
    cls_map = torch.randint(0, 19, (N, *hw), generator=g)
    feat = cents[cls_map].permute(0, 3, 1, 2).contiguous() * 0.6 + 0.7 * torch.randn((N, D, *hw), generator=g)
    pseudo_prob = synth.block_labels(g, N, HW[0], HW[1], block=8, ignore_frac=0.05)
    w = oc.centroid_weight(feat, cents)
    want, want_fp = oc.consensus_filter(w, pseudo_prob)
    up = ol.upsample_bilinear_ac(w, HW)
    top2 = up.topk(2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-5
    cf = mods["cc"].Class_Features(numbers=19, feat_dim=D)
    cf.objective_vectors = cents.to(DEV)
    assert_close(cf.get_centroid_weight(feat.to(DEV)), w, 3e-4, 1e-9, "weights")
    got, fp = cf.consensus_pseudo_labels(feat.to(DEV), pseudo_prob.to(DEV), return_feat_pseudo=True)
    assert bool((fp.cpu() == want_fp)[safe].all()) and bool((got.cpu() == want)[safe].all())
    assert float(safe.float().mean()) > 0.99


@pytest.mark.parametrize("N,D,hw", [(2, 256, (65, 129)), (1, 64, (7, 9)), (3, 100, (33, 31)), (1, 256, (16, 16))])
def test_class_sums_edge_shapes_vs_float64(mods, N, D, hw):
    """class_ids (plane loads eight at a time) + class_sums (a plane per block, four waves a quarter each, 16-deep rounds + tail): planes
    shorter than a round (63 pixels), exactly four rounds (256), ragged (8385), D not a multiple of anything -- per-class sums and counts
    against a float64 scatter."""
    g = synth.gen(N * 77 + D + hw[0])
    h, w = hw
    feat = torch.randn((N, D, h, w), generator=g)
    out = torch.randn((N, 19, h, w), generator=g)
    cf = mods["cc"].Class_Features(numbers=19, feat_dim=D)
    sums, counts = cf._class_sums(feat.to(DEV), out.to(DEV))[:2]
    arg = out.argmax(1)                                              # [N, h, w]
    want = torch.zeros((N, 19, D), dtype=torch.float64)
    want.scatter_add_(1, arg.reshape(N, -1, 1).expand(N, h * w, D), feat.double().permute(0, 2, 3, 1).reshape(N, h * w, D))
    wc = torch.zeros((N, 19), dtype=torch.int64).scatter_add_(1, arg.reshape(N, -1), torch.ones((N, h * w), dtype=torch.int64))
    assert torch.equal(counts.cpu().long(), wc)
    assert_close(sums, want, 2e-5, 2e-5, "class sums")


# ----------------------------------------------------------------------------- mIoU
def test_running_score(mods, golden):
    g = golden("miou")
    rs = mods["metrics"].runningScore(19, verbose=False)
    rs.update(g["gt"], g["pred"])
    assert np.array_equal(rs.confusion_matrix, g["hist"])
    sc, cls_iu = rs.get_scores()
    assert sc['Mean IoU : \t'] == pytest.approx(float(g["miou"]), rel=1e-12)
    rs.update(g.t("gt").to(DEV), g.t("pred").to(DEV))
    assert np.array_equal(rs.confusion_matrix, 2 * g["hist"])
    rs.reset()
    assert rs.confusion_matrix.sum() == 0
    assert np.array_equal(om.confusion(g["gt"], g["pred"]), g["hist"].astype(np.int64))
