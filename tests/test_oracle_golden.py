"""Pin the CPU oracle against captures of the reference (tests/golden/*.npz, made by
tools/gen_golden.py).  Runs on CPU; no GPU, no reference tree needed."""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle import centroids as oc
from oracle import classmix as ocm
from oracle import deeplab as od
from oracle import detweights, losses as ol, metrics as om, optim as oo, synth


# ----------------------------------------------------------------------------- losses
def test_kat1_cross_entropy(golden):
    g = golden("ce")
    x = g.t("kat_x").requires_grad_()
    loss = ol.cross_entropy2d(x, g.t("kat_y"))
    loss.backward()
    assert abs(float(loss) - 2.5379018784) < 2e-6
    assert_close(loss, g.t("kat_loss"), 1e-6, 0, "kat loss")
    assert_close(x.grad, g.t("kat_grad"), 1e-5, 1e-9, "kat grad")
    assert float(x.grad[0, 0, 0, 0]) == 0.0
    assert abs(float(x.grad[0, 0, 1, 0]) - 8.533773e-4) < 1e-9
    assert_close(ol.ce_grad(x.detach(), g.t("kat_y")), g.t("kat_grad"), 1e-5, 1e-9, "closed form")


def test_cross_entropy_random_and_all_ignored(golden):
    g = golden("ce")
    x = g.t("x").requires_grad_()
    loss = ol.cross_entropy2d(x, g.t("y"))
    loss.backward()
    assert_close(loss, g.t("loss"), 2e-6, 0, "loss")
    assert_close(x.grad, g.t("grad"), 1e-5, 1e-9, "grad")
    xe = g.t("allign_x").requires_grad_()
    le = ol.cross_entropy2d(xe, g.t("allign_y"))
    le.backward()
    assert float(le) == 0.0 and float(g.t("allign_loss")) == 0.0
    assert_close(xe.grad, g.t("allign_grad"), 0, 0, "all-ignored grad")


@pytest.mark.parametrize("tag,kw", [("a", dict(min_kept=50)), ("b", dict(min_kept=600)), ("c", dict(min_kept=100000)),
                                    ("d", dict(min_kept=300, thresh=0.5))])
def test_ohem_cross_entropy(golden, tag, kw):
    """OhemCrossEntropy of the reference in its regimes: threshold = thresh, threshold = k-th smallest probability,
    k clamped to n_valid - 1, low-res scores upsampled by the module."""
    g = golden("ohem")
    src = "b" if tag == "c" else tag
    x = g.t(src + "_x").clone().requires_grad_()
    loss, kept, thr = ol.ohem_cross_entropy(x, g.t(src + "_y"), **kw)
    loss.backward()
    assert_close(loss, g.t(tag + "_loss"), 1e-6, 1e-7, f"ohem {tag} loss")
    assert_close(x.grad, g.t(tag + "_grad"), 1e-5, 1e-9, f"ohem {tag} grad")
    if tag == "b":
        assert int(kept.sum()) == 600 and thr > 0.7          # exactly min_kept pixels lie below sorted[min_kept]


def test_kat2_distillation(golden):
    g = golden("distill")
    s = g.t("kat_s").requires_grad_()
    loss = ol.distillation_loss(g.t("kat_t"), s)
    loss.backward()
    assert abs(float(loss) - 4.6364974976) < 3e-6           # KAT-2
    assert_close(s.grad, g.t("kat_grad"), 1e-5, 1e-9, "kat grad")
    assert_close(ol.distill_grad(g.t("kat_t"), s.detach()), g.t("kat_grad"), 1e-5, 1e-9, "closed form")
    for scale, lk, gk in ((0.5, "loss", "grad"), (0.25, "loss_q", "grad_q")):
        s = g.t("s").clone().requires_grad_()
        loss = ol.distillation_loss(g.t("t"), s, scale)
        loss.backward()
        assert_close(loss, g.t(lk), 2e-6, 0, lk)
        assert_close(s.grad, g.t(gk), 1e-5, 1e-10, gk)


def test_upsample_and_fused_loss_block(golden):
    g = golden("upsample")
    y = ol.upsample_bilinear_ac(g.t("x"), (65, 97))
    assert_close(y, g.t("y"), 1e-5, 2e-6, "upsample")
    assert_close(F.interpolate(g.t("x"), (65, 97), mode="bilinear", align_corners=True), g.t("y"), 0, 0)
    stu = g.t("stu").requires_grad_()
    total, ce, di = ol.warmup_losses_lowres(stu, g.t("tea"), g.t("lab"))
    total.backward()
    assert_close(ce, g.t("ce"), 3e-6, 0, "ce")
    assert_close(di, g.t("distil"), 3e-6, 0, "distil")
    assert_close(total, g.t("total"), 3e-6, 0, "total")
    assert_close(stu.grad, g.t("grad_stu"), 2e-4, 2e-8, "grad wrt low-res student")


# ----------------------------------------------------------------------------- EMA / SGD
def test_kat4_ema(golden):
    g = golden("ema")
    its = g["its"].tolist()
    assert its == [0, 1, 2, 9, 998, 999, 1000, 5000]
    want = [0, 0.5, 2 / 3, 0.9, 1 - 1 / 999, 0.999, 0.999, 0.999]
    for i, a_ref, a_kat in zip(its, g["alphas"], want):
        assert oo.ema_alpha(i) == pytest.approx(a_ref, abs=0) and a_ref == pytest.approx(a_kat, abs=1e-12)
    stu = [g.t(k).clone() for k in ("s_a", "s_b", "s_c")]
    tea = [g.t(k).clone() for k in ("t_a", "t_b", "t_c")]
    for k, i in enumerate(its):
        for p in stu:
            p.add_(0.01 * (k + 1))
        oo.ema_update(tea, stu, i)
        for t, name in zip(tea, "abc"):
            assert_close(t, g.t(f"t_{name}_{k}"), 1e-6, 1e-7, f"ema it={i} {name}")
    assert bool(g["created_equal"][0]) and bool(g["created_buf_untouched"][0])
    assert np.array_equal(g["t_buf"], g["t_buf_after"])      # buffers are not EMA'd


def test_sgd_duplicates(golden):
    g = golden("sgd")
    mults, groups = g["mults"].tolist(), g["groups"].tolist()
    params = [g.t(f"p0_{i}").clone() for i in range(4)]
    bufs = [torch.zeros_like(p) for p in params]
    for step in range(3):
        lr1, lr10 = g[f"lr_{step}"].tolist()
        assert lr1 == pytest.approx(oo.poly_lr(2.5e-4, step, 100, 0.9), rel=1e-12)
        assert lr10 == pytest.approx(10 * lr1, rel=1e-12)
        lrs = [lr10 if gr else lr1 for gr in groups]
        grads = [g.t(f"g{step}_{i}") for i in range(4)]
        oo.sgd_step_dup(params, grads, bufs, mults, lrs, first_step=(step == 0))
        for i in range(4):
            assert_close(params[i], g.t(f"p{step + 1}_{i}"), 1e-6, 1e-7, f"param {i} step {step}")
            assert_close(bufs[i], g.t(f"buf{step + 1}_{i}"), 1e-6, 1e-7, f"buf {i} step {step}")
    # scalar known answer of SURVEY App. A-9
    p, b = [torch.ones(1)], [torch.zeros(1)]
    oo.sgd_step_dup(p, [torch.full((1,), 2.0)], b, [3], [0.1], 0.9, 0.01, first_step=True)
    assert float(p[0]) == pytest.approx(0.39760, abs=2e-5)
    assert float(p[0]) == pytest.approx(float(g["kat_scalar"][0]), rel=1e-6)
    oo.sgd_step_dup(p, [torch.full((1,), 2.0)], b, [3], [0.1], 0.9, 0.01, first_step=False)
    assert float(p[0]) == pytest.approx(-1.21424, abs=2e-5)
    assert float(p[0]) == pytest.approx(float(g["kat_scalar"][1]), rel=1e-6)


# ----------------------------------------------------------------------------- ClassMix
def test_kat6_and_classmix(golden):
    g = golden("classmix")
    random.seed(0)
    assert random.sample([0, 1, 2, 5, 8, 10, 13, 255], 4) == [13, 255, 5, 0] == g["kat6"].tolist()
    labels, bg, fg, bg_lab = g.t("labels"), g.t("bg"), g.t("fg"), g.t("bg_lab")
    random.seed(int(g["seed"][0]))
    mixed, mask, sels, mixed_lab = ocm.classmix(bg, fg, labels, random, bg_labels=bg_lab)
    for b, sel in enumerate(sels):
        want = [v for v in g["sel"][b].tolist() if v >= 0]
        assert sel == want
    assert torch.equal(mask, g.t("mask"))
    assert torch.equal(mixed, g.t("mixed"))                  # bit-exact: pure select arithmetic
    assert torch.equal(mixed_lab, g.t("mixed_lab"))
    assert 255 in sels[1] and not bool((labels[1] == 255).any())   # 255 force-added when absent


# ----------------------------------------------------------------------------- centroids
def test_kat3_centroid_weight(golden):
    g = golden("centroid")
    w = oc.centroid_weight(g.t("kat_f"), g.t("kat_c"))
    assert_close(w, g.t("kat_w"), 2e-4, 1e-12, "kat weights")
    kat = [1.3002333e-06, 3.7265915e-09, 5.8784060e-02, 2.1599791e-10]
    assert_close(w[0, :4, 0, 0], torch.tensor(kat), 3e-4, 0, "KAT-3 numbers")
    assert bool((w.argmax(1) == 14).all())
    _, arg = oc.consensus_filter(w, torch.zeros((1, 5, 5), dtype=torch.int64))
    assert torch.equal(arg, g.t("kat_arg")) and bool((arg == 14).all())


def test_centroid_weights_and_consensus(golden):
    g = golden("centroid")
    feat, cents = g.t("feat"), g.t("cents")
    assert_close(-oc.centroid_distance(feat, cents), g.t("neg_dist"), 1e-5, 1e-5, "distance")
    w = oc.centroid_weight(feat, cents)
    assert_close(w, g.t("w"), 2e-4, 1e-9, "weights")
    pseudo, feat_pseudo = oc.consensus_filter(w, g.t("pseudo_prob"))
    safe = g.t("margin") > 1e-5                               # away from argmax near-ties
    assert bool((feat_pseudo == g.t("feat_pseudo"))[safe].all())
    assert bool((pseudo == g.t("pseudo"))[safe].all())
    assert float(safe.float().mean()) > 0.999
    kept = float((g.t("pseudo") != 255).float().mean())
    assert 0.2 < kept < 0.9                                   # the fixture exercises both branches


def test_kat5_class_means_and_sequential_ema(golden):
    g = golden("meanvec")
    feat, out, cents = g.t("feat"), g.t("out"), g.t("cents")
    for tag, lab in (("nolab", None), ("lab", g.t("lab_lr")[:, 0])):
        vecs, ids, _ = oc.class_mean_vectors(feat, out, lab)
        assert ids == g[f"{tag}_ids"].tolist()
        assert_close(torch.stack(vecs), g.t(f"{tag}_vecs"), 1e-5, 1e-6, f"{tag} vectors")
        c, n = cents.clone(), torch.zeros(19)
        oc.centroid_ema_apply(c, n, vecs, ids)
        assert_close(c, g.t(f"{tag}_cents"), 1e-6, 1e-7, f"{tag} centroids")
        assert torch.equal(n, g.t(f"{tag}_nums"))
    assert len(g["lab_ids"]) < len(g["nolab_ids"])            # label consensus drops classes
    assert set(g["lab_ids"].tolist()) <= set(g["nolab_ids"].tolist())
    vecs, ids, _ = oc.class_mean_vectors(feat, out, None)
    c, n = torch.zeros(19, 256), torch.zeros(19)
    for _ in range(2):
        oc.centroid_mean_apply(c, n, vecs, ids)
    assert_close(c, g.t("mean_cents"), 1e-5, 1e-6, "mean-mode centroids")
    assert torch.equal(n, g.t("mean_nums"))
    near = oc.nearest_downsample_labels(g.t("full_lab"), (17, 17))
    assert torch.equal(near.float(), g.t("near_lr")[:, 0])
    # KAT-5 closed forms: 3x3 map, 9 distinct classes -> nothing passes the >=5 px rule
    f = torch.randn(1, 256, 3, 3)
    o = torch.zeros(1, 19, 3, 3)
    o[0, torch.arange(9), torch.arange(9) // 3, torch.arange(9) % 3] = 1.0
    assert oc.class_mean_vectors(f, o)[1] == []
    f9 = torch.randn(1, 256, 9, 9)
    o9 = torch.zeros(1, 19, 9, 9)
    o9[:, 4] = 1.0
    v, ids, _ = oc.class_mean_vectors(f9, o9)
    assert ids == [4]
    assert_close(v[0], f9.mean(dim=(2, 3))[0], 1e-5, 1e-6)


# ----------------------------------------------------------------------------- mIoU
def test_miou(golden):
    g = golden("miou")
    hist = om.confusion(g["gt"], g["pred"])
    assert np.array_equal(hist, g["hist"].astype(np.int64))
    sc = om.scores(hist)
    assert sc["miou"] == pytest.approx(float(g["miou"]), rel=1e-12)
    assert sc["acc"] == pytest.approx(float(g["acc"]), rel=1e-12)
    assert sc["acc_cls"] == pytest.approx(float(g["acc_cls"]), rel=1e-12)
    assert sc["fwavacc"] == pytest.approx(float(g["fwavacc"]), rel=1e-12)
    assert np.allclose(sc["iu"], g["iu"], rtol=1e-12, equal_nan=True)


# ----------------------------------------------------------------------------- model
def _aspp64_sd():
    arch = od.Arch(layers=(1, 1, 1, 1), planes=(4, 4, 4, 16))     # head input = 16*4 = 64
    shapes = {k: v for k, v in od.state_shapes(arch).items() if k.startswith("final.")}
    sd = {}
    for k, (shp, kind) in shapes.items():
        ref_kind = {"head": "conv"}.get(kind, kind)               # reference fill treats all 4-D as conv
        sd[k] = detweights.fill("aspp64." + k[len("final."):], shp, ref_kind)
    return arch, sd


def test_aspp_head(golden):
    g = golden("aspp")
    arch, sd = _aspp64_sd()
    for v in sd.values():
        v.requires_grad_()
    x = g.t("x").requires_grad_()
    out, feat = od.aspp_head(sd, x, arch, keep_mask=None)
    assert_close(out, g.t("out"), 1e-4, 1e-5, "out")
    assert_close(feat, g.t("feat"), 1e-4, 1e-5, "feat")
    ((out * g.t("probe")).sum() + (feat * g.t("probe_f")).sum()).backward()
    assert_close(x.grad, g.t("gx"), 1e-3, 1e-5, "grad x")
    for k, v in sd.items():
        gk = "gw_" + k[len("final."):].replace(".", "_")
        if gk in g:
            ref = g.t(gk)
            assert_close(v.grad, ref, 2e-3, 1e-4 * float(ref.abs().max()) + 1e-7, gk)
        else:
            cs, l1 = g[gk + "__sum"].tolist()
            ref = g.t(gk + "__sample")
            assert float(v.grad.abs().sum()) == pytest.approx(l1, rel=1e-3), gk
            assert_close(v.grad.reshape(-1)[::97], ref, 2e-3, 1e-4 * float(ref.abs().max()) + 1e-7, gk)


def test_aspp_head_live_dropout(golden):
    """The oracle's `keep_mask` against the reference head with nn.Dropout2d(0.1) LIVE (tests/golden/aspp_dropout.npz: the drawn
    mask read back from the reference's own `feat`): scale 1 / (1 - p) on kept (image, channel) pairs, zero on dropped ones, `feat` is
    the dropped tensor, gradients flow through the mask."""
    g = golden("aspp_dropout")
    arch, sd = _aspp64_sd()
    for v in sd.values():
        v.requires_grad_()
    x = g.t("x").requires_grad_()
    keep = g.t("keep")
    assert arch.droprate == pytest.approx(0.1) and 0 < int((1 - keep).sum()) < keep.numel() // 4
    out, feat = od.aspp_head(sd, x, arch, keep_mask=keep)
    assert_close(out, g.t("out"), 1e-4, 1e-5, "out")
    assert_close(feat, g.t("feat"), 1e-4, 1e-5, "feat")
    assert bool((feat.detach().abs().amax(dim=(2, 3)) > 0).float().eq(keep).all())
    ((out * g.t("probe")).sum() + (feat * g.t("probe_f")).sum()).backward()
    assert_close(x.grad, g.t("gx"), 1e-3, 1e-5, "grad x")
    for k, v in sd.items():
        gk = "gw_" + k[len("final."):].replace(".", "_")
        if gk in g:
            ref = g.t(gk)
            assert_close(v.grad, ref, 2e-3, 1e-4 * float(ref.abs().max()) + 1e-7, gk)
        else:
            cs, l1 = g[gk + "__sum"].tolist()
            ref = g.t(gk + "__sample")
            assert float(v.grad.abs().sum()) == pytest.approx(l1, rel=1e-3), gk
            assert_close(v.grad.reshape(-1)[::97], ref, 2e-3, 1e-4 * float(ref.abs().max()) + 1e-7, gk)


@pytest.mark.timeout(1200)
def test_oracle_training_run_reaches_the_reference_curve(golden):
    """The oracle's whole training loop -- oracle/step.py::Trainer.warmup_step with Dropout2d LIVE (its own draws), ClassMix, EMA teacher,
    duplicate-aware SGD -- run for the first 100 steps of the trained-model experiment of tests/golden/trainmiou.npz (seed 0's data and
    ClassMix draws), then the reference's two-scale validation restated (oracle/evaluate.py) on the 32 held-out images: the reference's three
    training runs score 66.09 / 66.20 / 66.04 % mIoU at that checkpoint.  That agreement is a coincidence of those three dropout streams:
    the curve climbs 0.2-0.6 points per step there (37.9 at step 50, 66.1 at 100, 76.6 at 150), single seeds of the HIP path land between 64
    and 69, and the oracle with its own draws at 71.1 -- so the bound is +-8 points around the reference's mean plus the training loss of
    steps 50..99 within 10 % (measured: 1.322 vs 1.382).  A training-level defect of the restatement (a wrong gradient scale, an EMA or
    BatchNorm slip, dropout on the wrong tensor) leaves the curve tens of points lower.  ~2 minutes on 8 cores: the one long CPU test."""
    import random
    from oracle import evaluate as oe
    from oracle import metrics as om
    from oracle import step as ost
    g = golden("trainmiou")
    B, H, W, steps, block, every, n_val, data_seed0, val_seed0 = (int(v) for v in g["geometry"])
    ref100 = 100.0 * np.asarray(g["curve"], dtype=np.float64)[:, 1]                # checkpoint at step 100, per seed
    assert every == 50
    seed = 0
    gm = synth.gen(1234 + seed)
    sd_s, sd_t = detweights.state_dict(od.RESNET101), detweights.state_dict(od.RESNET101)
    tr = ost.Trainer(sd_s, sd_t, arch=od.RESNET101, base_lr=float(g["lr"]), droprate_off=False,
                     keep_masks=lambda role, n, w: (torch.rand((n, w), generator=gm) >= od.RESNET101.droprate).float())
    rng = random.Random()
    rng.seed(4321 + seed)
    ce = []
    for it in range(100):
        batch = synth.learnable_batch(data_seed0 + 1000 * seed + it, B, H, W, block=block)
        ce.append(tr.warmup_step(it, *batch, rng)["ce"])
    hist = np.zeros((19, 19), dtype=np.int64)
    with torch.no_grad():
        for i in range(n_val):
            img, _, _, gt = synth.learnable_batch(val_seed0 + i, 1, H, W, block=block)
            _, h_i, _ = oe.evaluate_two_scale(lambda x: od.forward(tr.s, x, od.RESNET101, training=False)[2], img, gt)
            hist += h_i
    miou = 100.0 * float(om.scores(hist.astype(np.float64))["miou"])
    ref_ce = np.asarray(g["ce"], dtype=np.float64)[:, 50:100].mean()
    print(f"oracle after 100 steps: val mIoU {miou:.2f} (reference {ref100.tolist()}), CE of steps 50..99 {np.mean(ce[50:]):.4f} (reference {ref_ce:.4f})")
    assert abs(miou - float(ref100.mean())) <= 8.0, (miou, ref100.tolist())
    assert np.mean(ce[50:]) == pytest.approx(ref_ce, rel=0.1)


def test_model_structure_matches_reference(golden):
    g = golden("model")
    shapes = od.state_shapes(od.RESNET101)
    assert list(shapes.keys()) == g["state_keys"].tolist()
    pkeys = [k for k, (_, kind) in shapes.items() if kind not in ("bn_rm", "bn_rv", "bn_nbt")]
    assert pkeys == g["param_keys"].tolist()
    numel = sum(int(np.prod(s)) for k, (s, kind) in shapes.items() if k in set(pkeys))
    assert numel == int(g["numel_params"]) == 65063568
    train = sum(int(np.prod(s)) for k, (s, kind) in shapes.items()
                if k in set(pkeys) and kind not in ("bn_w", "bn_b"))
    assert train == int(g["numel_trainable"]) == 64958224
    assert int(g["n_params"]) == 341 and int(g["g1_entries"]) == 315 and int(g["g1_unique"]) == 104
    assert g["g1_mult_hist"].tolist() == [0, 0, 1, 99, 4, 0]
    assert (int(g["g1_mult_stem"]), int(g["g1_mult_block"]), int(g["g1_mult_down"])) == (2, 3, 4)
    assert int(g["g10_entries"]) == 29


@pytest.mark.timeout(600)
def test_model_forward_eval_and_train(golden):
    g = golden("model")
    sd = detweights.state_dict(od.RESNET101)
    x = g.t("x")
    with torch.no_grad():
        sh, dp, out, feat = od.forward(sd, x, training=False)
    assert list(sh.shape) == g["shallow_shape"].tolist() and list(dp.shape) == g["deep_shape"].tolist()
    assert_close(out, g.t("out_eval"), 1e-4, 1e-5, "eval logits")
    assert_close(feat, g.t("feat_eval"), 1e-4, 1e-5, "eval feat")
    assert synth.checksum(sh) == pytest.approx(float(g["shallow_sum"]), rel=1e-4, abs=1e-3)
    assert synth.checksum(dp) == pytest.approx(float(g["deep_sum"]), rel=1e-4, abs=1e-3)
    for k, (_, kind) in od.state_shapes().items():
        if kind in ("conv", "bias", "gn_w", "gn_b", "lin", "head"):
            sd[k].requires_grad_()
    keep = torch.ones(2, 256)                                   # dropout forced off: keep all, scale 1/(1-p)
    arch0 = od.Arch(droprate=0.0)
    _, _, out, feat = od.forward(sd, x, arch0, training=True, keep_mask=keep, update_stats=True)
    assert_close(out, g.t("out_train"), 2e-4, 2e-5, "train logits")
    (out * g.t("probe")).sum().backward()
    assert_close(sd["final.head.1.weight"].grad, g.t("g_head"), 1e-3, 1e-4, "head grad")
    for n in ["layer0.0.weight", "layer1.0.conv1.weight", "layer2.3.conv2.weight",
              "layer3.22.conv3.weight", "layer4.0.downsample.0.weight",
              "final.conv2d_list.3.0.weight", "final.conv2d_list.0.1.weight",
              "final.bottleneck.0.se.0.weight", "final.bottleneck.1.bias"]:
        cs, l1 = g["g_" + n.replace(".", "_")].tolist()
        assert float(sd[n].grad.abs().sum()) == pytest.approx(l1, rel=2e-3), n
    assert_close(sd["layer1.0.bn1.running_mean"], g.t("rm_after"), 1e-5, 1e-6, "running mean")
    assert_close(sd["layer4.2.bn3.running_var"], g.t("rv_after"), 1e-4, 1e-6, "running var")


def test_model_forward_benchmark_geometry(golden):
    """The oracle at the 512x1024 benchmark geometry (65x129 map) against the capture of the reference: train-mode
    forward (batch-statistics BN), the size bench.py's cpu_baseline leg and the full-size GPU tests lean on."""
    g = golden("full512x1024")
    _, H, W = (int(v) for v in g["geometry"])
    gen = synth.gen(int(g["seed"]))
    x = torch.rand((2, 3, H, W), generator=gen) * 2 - 1
    sd = detweights.state_dict(od.RESNET101)
    with torch.no_grad():
        _, dp, out, feat = od.forward(sd, x, od.Arch(droprate=0.0), training=True, keep_mask=torch.ones(2, 256))
    want = g.t("out")
    assert float((out - want).abs().max()) < 2e-4 * float(want.abs().max())
    assert float(feat.abs().sum()) == pytest.approx(float(g["feat_sum"][1]), rel=1e-5)
    assert float(dp.abs().sum()) == pytest.approx(float(g["deep_sum"][1]), rel=1e-5)


# ----------------------------------------------------------------------------- whole warm-up step
@pytest.mark.timeout(900)
def test_warmup_three_steps(golden):
    from oracle import step as ost
    g = golden("step")
    tr = ost.Trainer(detweights.state_dict(), detweights.state_dict())
    random.seed(77)
    for it in range(3):
        x, x_aug, rec, lab = synth.warmup_batch(1000 + it, 2, 128, 128, block=16)
        log = tr.warmup_step(it, x, x_aug, rec, lab, random)
        assert log["ce"] == pytest.approx(float(g["ce"][it]), rel=2e-4), it
        assert log["distil"] == pytest.approx(float(g["distil"][it]), rel=2e-4), it
        assert log["lr"] == pytest.approx(float(g["lr"][it]), rel=1e-12)
    assert_close(tr.s["final.head.1.weight"], g.t("student_head"), 1e-3, 1e-6, "student head")
    assert_close(tr.t["final.head.1.weight"], g.t("teacher_head"), 1e-3, 1e-6, "teacher head")
    assert_close(tr.s["layer1.0.bn1.running_mean"], g.t("stu_rm"), 1e-4, 1e-6, "student running mean")
    assert_close(tr.t["layer1.0.bn1.running_mean"], g.t("tea_rm"), 1e-4, 1e-6, "teacher running mean")
    for n in ["layer0.0.weight", "layer3.10.conv2.weight", "final.conv2d_list.2.0.weight",
              "layer2.0.downsample.0.weight"]:
        key = n.replace(".", "_")
        assert synth.checksum(tr.s[n]) == pytest.approx(float(g["ps_" + key]), rel=1e-3, abs=1e-4), n
        assert synth.checksum(tr.t[n]) == pytest.approx(float(g["pt_" + key]), rel=1e-3, abs=1e-4), n
    xp = synth.warmup_batch(2000, 1, 128, 128, block=16)[0]
    with torch.no_grad():
        so = od.forward(tr.s, xp, training=False)[2]
        to = od.forward(tr.t, xp, training=False)[2]
    assert_close(so, g.t("probe_student"), 2e-3, 2e-4, "student probe logits")
    assert_close(to, g.t("probe_teacher"), 2e-3, 2e-4, "teacher probe logits")


@pytest.mark.timeout(900)
def test_warmup_trajectory_first_steps(golden):
    """The oracle's step against the 25-step capture of the reference's loop (tests/golden/traj25.npz): the first six steps, each
    loss within the capture's own rounding floor (two runs of the reference that differ in summation order: `floor_*`)."""
    from oracle import step as ost
    g = golden("traj25")
    B, H, W, steps, seed0, block, mix_seed = (int(v) for v in g["geometry"])
    tr = ost.Trainer(detweights.state_dict(), detweights.state_dict())
    random.seed(mix_seed)
    floor = max(float(g["floor_ce_dev"].max()), float(g["floor_distil_dev"].max()))
    assert 1e-6 < floor < 1e-3 and steps == 25
    for it in range(6):
        x, x_aug, rec, lab = synth.warmup_batch(seed0 + it, B, H, W, block=block)
        log = tr.warmup_step(it, x, x_aug, rec, lab, random)
        assert log["ce"] == pytest.approx(float(g["ce"][it]), rel=3 * floor), it
        assert log["distil"] == pytest.approx(float(g["distil"][it]), rel=3 * floor), it
        assert log["lr"] == pytest.approx(float(g["lr"][it]), rel=1e-12)
