#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (read-only at /root/reference)
on PyTorch-CPU in the build container (SURVEY.md App. C recipe).

Only inputs and the reference's outputs are stored; no reference source travels.
Run from the repo root:  python tools/gen_golden.py [names...]
"""
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/domain_adaptation/GTA5"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

# ---- import recipe: stub the absent third-party modules, neutralise .cuda() and the download
tv = types.ModuleType("torchvision")
tv.models = types.ModuleType("torchvision.models")
tv.utils = types.ModuleType("torchvision.utils")
sys.modules.update({"torchvision": tv, "torchvision.models": tv.models,
                    "torchvision.utils": tv.utils, "kornia": types.ModuleType("kornia")})
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
import torch.utils.model_zoo as mz  # noqa: E402

mz.load_url = lambda *a, **k: {}
import warnings  # noqa: E402

warnings.filterwarnings("ignore")

from model.model_noaux import SegModel  # noqa: E402  (reference)
from model.seg_model_noaux import Classifier_Module2  # noqa: E402  (reference)
from util.loss import OhemCrossEntropy, cross_entropy2d, distillation_loss  # noqa: E402  (reference)
from util.utils import (adjust_learning_rate, create_teacher_params,  # noqa: E402
                        update_teacher_params)
from util.metrics import runningScore  # noqa: E402  (reference)
from calc_centroids import Class_Features  # noqa: E402  (reference)

from oracle import detweights, synth  # noqa: E402  (build-owned generators only)
from oracle.deeplab import RESNET101, Arch  # noqa: E402


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def striped_labels(g, shape, n_classes=19):
    lab = torch.randint(0, n_classes, shape, generator=g)
    lab[:, ::5, :] = 255
    lab[:, :, 3::11] = 255
    return lab


# ------------------------------------------------------------------ G-ce (KAT-1)
def gen_ce():
    x = torch.sin(0.01 * torch.arange(608, dtype=torch.float32)).reshape(2, 19, 4, 4).requires_grad_()
    y = (torch.arange(32) % 19).reshape(2, 4, 4)
    y[0, 0, :] = 255
    loss = cross_entropy2d(x, y)
    loss.backward()
    g = synth.gen(0)
    xr = (3.0 * torch.randn((2, 19, 33, 33), generator=g)).requires_grad_()
    yr = striped_labels(g, (2, 33, 33))
    lr = cross_entropy2d(xr, yr)
    lr.backward()
    # all-ignored and no-ignored edge cases
    xe = torch.randn((1, 19, 5, 7), generator=g).requires_grad_()
    ye = torch.full((1, 5, 7), 255, dtype=torch.int64)
    le = cross_entropy2d(xe, ye)
    le.backward()
    save("ce", kat_x=x, kat_y=y, kat_loss=loss, kat_grad=x.grad,
         x=xr, y=yr, loss=lr, grad=xr.grad,
         allign_x=xe, allign_y=ye, allign_loss=le, allign_grad=xe.grad)


# ------------------------------------------------------------------ G-distil (KAT-2)
def gen_distill():
    a = torch.arange(1216, dtype=torch.float32)
    t = torch.cos(0.01 * a).reshape(4, 19, 4, 4)
    s = torch.sin(0.01 * a).reshape(4, 19, 4, 4).requires_grad_()
    loss = distillation_loss(t, s)
    loss.backward()
    g = synth.gen(1)
    tr = 2.0 * torch.randn((4, 19, 33, 33), generator=g)
    sr = (2.0 * torch.randn((4, 19, 33, 33), generator=g)).requires_grad_()
    lr = distillation_loss(tr, sr)
    lr.backward()
    sq = sr.detach().clone().requires_grad_()
    lq = distillation_loss(tr, sq, scale=0.25)
    lq.backward()
    save("distill", kat_t=t, kat_s=s, kat_loss=loss, kat_grad=s.grad,
         t=tr, s=sr, loss=lr, grad=sr.grad, loss_q=lq, grad_q=sq.grad)


# ------------------------------------------------------------------ upsample + fused loss block
def gen_upsample():
    g = synth.gen(2)
    x = torch.randn((2, 19, 9, 13), generator=g)
    up = torch.nn.Upsample(size=[65, 97], mode="bilinear", align_corners=True)
    y = up(x)
    # warm-up loss block at the low-res boundary: warm_up.py:267-282,299 re-enacted
    B = 2
    stu = (2.0 * torch.randn((2 * B, 19, 9, 13), generator=g)).requires_grad_()
    tea = 2.0 * torch.randn((2 * B, 19, 9, 13), generator=g)
    lab = striped_labels(g, (B, 65, 97))
    s_up, t_up = up(stu), up(tea)
    ce = cross_entropy2d(s_up[:B], lab)
    di = distillation_loss(t_up, s_up)
    total = 1.0 * ce + 0.5 * di
    total.backward()
    save("upsample", x=x, y=y, stu=stu, tea=tea, lab=lab, ce=ce, distil=di, total=total,
         grad_stu=stu.grad)


# ------------------------------------------------------------------ G-ema (KAT-4)
def gen_ema():
    class Net(torch.nn.Module):
        def __init__(self, seed):
            super().__init__()
            g = synth.gen(seed)
            self.a = torch.nn.Parameter(torch.randn((7, 5), generator=g))
            self.b = torch.nn.Parameter(torch.randn((1031,), generator=g))
            self.c = torch.nn.Parameter(torch.randn((3, 4, 3, 3), generator=g))
            self.register_buffer("buf", torch.randn((4,), generator=g))

    its = [0, 1, 2, 9, 998, 999, 1000, 5000]
    alphas = [min(1 - 1 / (i + 1), 0.999) for i in its]
    out = {"its": np.array(its), "alphas": np.array(alphas, dtype=np.float64)}
    stu, tea = Net(10), Net(11)
    out.update(s_a=stu.a.detach().clone(), s_b=stu.b.detach().clone(), s_c=stu.c.detach().clone(),
               t_a=tea.a.detach().clone(), t_b=tea.b.detach().clone(), t_c=tea.c.detach().clone(),
               t_buf=tea.buf.clone())
    # sequence: before update k the student moves by +0.01*(k+1) (stand-in for an SGD step)
    for k, i in enumerate(its):
        with torch.no_grad():
            for p in stu.parameters():
                p.add_(0.01 * (k + 1))
            update_teacher_params(tea, stu, i)
        out.update({f"t_a_{k}": tea.a.detach().clone(), f"t_b_{k}": tea.b.detach().clone(),
                    f"t_c_{k}": tea.c.detach().clone()})
    out["t_buf_after"] = tea.buf.clone()
    tea2 = Net(12)
    buf_before = tea2.buf.clone()
    create_teacher_params(tea2, stu)
    out.update(created_equal=np.array([bool(torch.equal(tea2.a, stu.a) and torch.equal(tea2.b, stu.b)
                                            and torch.equal(tea2.c, stu.c))]),
               created_buf_untouched=np.array([bool(torch.equal(tea2.buf, buf_before))]))
    save("ema", **out)


# ------------------------------------------------------------------ G-sgd (App. A-9)
def gen_sgd():
    g = synth.gen(3)
    shapes = [(5, 3), (17,), (2, 3, 3, 3), (9,)]
    mults = [1, 3, 4, 2]
    groups = [0, 0, 0, 1]                       # last tensor sits in the 10x group
    p0 = [torch.randn(s, generator=g) for s in shapes]
    params = [torch.nn.Parameter(p.clone()) for p in p0]
    g1x = [p for p, m, gr in zip(params, mults, groups) if gr == 0 for _ in range(m)]
    g10x = [p for p, m, gr in zip(params, mults, groups) if gr == 1 for _ in range(m)]
    opt = torch.optim.SGD([{"params": g1x, "lr": 2.5e-4}, {"params": g10x, "lr": 2.5e-3}],
                          lr=2.5e-4, momentum=0.9, weight_decay=5e-4)
    out = {"mults": np.array(mults), "groups": np.array(groups)}
    for i, p in enumerate(p0):
        out[f"p0_{i}"] = p
    for step in range(3):
        adjust_learning_rate([opt], base_lr=2.5e-4, i_iter=step, max_iter=100, power=0.9)
        out[f"lr_{step}"] = np.array([opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]])
        for i, p in enumerate(params):
            p.grad = torch.randn(p.shape, generator=g)
            out[f"g{step}_{i}"] = p.grad.clone()
        opt.step()
        for i, p in enumerate(params):
            out[f"p{step + 1}_{i}"] = p.detach().clone()
            out[f"buf{step + 1}_{i}"] = opt.state[p]["momentum_buffer"].clone()
    # scalar known-answer of App. A-9: p=1,g=2,lr=.1,wd=.01,k=3
    q = torch.nn.Parameter(torch.ones(1))
    o2 = torch.optim.SGD([{"params": [q, q, q]}], lr=0.1, momentum=0.9, weight_decay=0.01)
    ka = []
    for _ in range(2):
        q.grad = torch.full((1,), 2.0)
        o2.step()
        ka.append(float(q))
    out["kat_scalar"] = np.array(ka)
    save("sgd", **out)


# ------------------------------------------------------------------ G-classmix (KAT-6)
def gen_classmix():
    g = synth.gen(4)
    B, H, W = 2, 64, 64
    labels = synth.block_labels(g, B, H, W, block=8, ignore_frac=0.02)
    labels[1][labels[1] == 255] = 3            # image 1 has no ignore pixels
    bg = torch.randn((B, 3, H, W), generator=g)        # translated image / aug target
    fg = torch.randn((B, 3, H, W), generator=g)        # aug source
    bg_lab = synth.block_labels(g, B, H, W, block=16, ignore_frac=0.1)   # filtered pseudo-labels
    random.seed(0)
    kat6 = random.sample([0, 1, 2, 5, 8, 10, 13, 255], 4)
    # re-enactment of the inline block (warm_up.py:240-259 / self_training.py:306-325)
    random.seed(1234)
    mask = torch.zeros(labels.size())
    mixed_lab = bg_lab.clone()
    chosen = []
    for i in range(B):
        present = torch.unique(labels[i]).tolist()
        pick = random.sample(present, len(present) // 2)
        if 255 not in pick:
            pick.append(255)
        chosen.append(pick)
        for c in pick:
            mixed_lab[i][labels[i] == c] = c
            mask[i][labels[i] == c] = 1
    mixed = torch.zeros(bg.size())
    for i in range(B):
        mixed[i] = torch.mul(bg[i], 1 - mask[i]) + torch.mul(fg[i], mask[i])
    sel = np.full((B, 24), -1, dtype=np.int64)
    for i, pick in enumerate(chosen):
        sel[i, :len(pick)] = pick
    save("classmix", labels=labels, bg=bg, fg=fg, bg_lab=bg_lab, kat6=np.array(kat6),
         seed=np.array([1234]), sel=sel, mask=mask, mixed=mixed, mixed_lab=mixed_lab.long())


# ------------------------------------------------------------------ G-centroid (KAT-3)
def gen_centroid():
    cf = Class_Features(numbers=19)
    f = torch.sin(0.001 * torch.arange(2304, dtype=torch.float32)).reshape(1, 256, 3, 3)
    c = torch.cos(0.01 * torch.arange(4864, dtype=torch.float32)).reshape(19, 256)
    cf.objective_vectors = c.clone()
    w_kat = cf.get_centroid_weight(f)
    up5 = torch.nn.Upsample(size=[5, 5], mode="bilinear", align_corners=True)
    kat_arg = up5(w_kat).max(1, keepdim=True)[1].squeeze(1)
    g = synth.gen(5)
    cents = torch.randn((19, 256), generator=g)
    # features clustered around centroids so that weights are not degenerate
    cls_map = torch.randint(0, 19, (2, 17, 17), generator=g)
    feat = cents[cls_map].permute(0, 3, 1, 2).contiguous() * 0.6 + 0.7 * torch.randn((2, 256, 17, 17), generator=g)
    cf.objective_vectors = cents.clone()
    w = cf.get_centroid_weight(feat)
    dist = cf.get_centroid_distance(feat)
    up = torch.nn.Upsample(size=[128, 128], mode="bilinear", align_corners=True)
    pseudo_prob = synth.block_labels(g, 2, 128, 128, block=8, ignore_frac=0.05)
    # agree with the centroid label on ~half of the pixels
    wup = up(w)
    feat_pseudo = wup.max(1, keepdim=True)[1].squeeze(1)
    agree = torch.rand((2, 128, 128), generator=g) < 0.5
    pseudo_prob = torch.where(agree, feat_pseudo, pseudo_prob)
    pseudo = pseudo_prob.clone()
    pseudo[pseudo_prob != feat_pseudo] = 255
    top2 = wup.topk(2, dim=1)[0]
    save("centroid", kat_f=f, kat_c=c, kat_w=w_kat, kat_arg=kat_arg,
         feat=feat, cents=cents, w=w, neg_dist=dist, pseudo_prob=pseudo_prob,
         feat_pseudo=feat_pseudo, pseudo=pseudo, margin=(top2[:, 0] - top2[:, 1]))


# ------------------------------------------------------------------ G-meanvec (KAT-5)
def gen_meanvec():
    g = synth.gen(6)
    cf = Class_Features(numbers=19)
    cents = torch.randn((19, 256), generator=g)
    cf.objective_vectors = cents.clone()
    cf.objective_vectors_num = torch.zeros([19])
    N, h, w = 2, 17, 17
    feat = torch.randn((N, 256, h, w), generator=g)
    out = torch.randn((N, 19, h, w), generator=g)
    # coarse class structure so that several classes pass the >=5 px rule, some don't
    blocks = torch.randint(0, 19, (N, 5, 5), generator=g).repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :h, :w]
    out.scatter_add_(1, blocks[:, None], torch.full((N, 1, h, w), 6.0))
    full_lab = synth.block_labels(g, N, 128, 128, block=16, ignore_frac=0.05)
    argm = out.argmax(1)
    lab_lr = F.interpolate(full_lab.reshape(N, 1, 128, 128).float(), size=(h, w), mode="nearest")
    take = torch.rand((N, 1, h, w), generator=g) < 0.3
    lab_lr = torch.where(take, argm[:, None].float(), lab_lr)
    res = {}
    for tag, labels in (("nolab", None), ("lab", lab_lr)):
        vecs, ids = cf.calculate_mean_vector(feat, out, labels)
        res[f"{tag}_ids"] = np.array(ids, dtype=np.int64)
        res[f"{tag}_vecs"] = (torch.stack([v.reshape(256) for v in vecs]) if vecs
                              else torch.zeros((0, 256)))
        cf.objective_vectors = cents.clone()
        cf.objective_vectors_num = torch.zeros([19])
        for t in range(len(ids)):
            cf.update_objective_SingleVector(ids[t], vecs[t].detach(), start_mean=False)
        res[f"{tag}_cents"] = cf.objective_vectors.clone()
        res[f"{tag}_nums"] = cf.objective_vectors_num.clone()
    # 'mean' mode of the offline pass (calc_centroids.py:78-79)
    cf.objective_vectors = torch.zeros([19, 256])
    cf.objective_vectors_num = torch.zeros([19])
    vecs, ids = cf.calculate_mean_vector(feat, out)
    for rep in range(2):
        for t in range(len(ids)):
            cf.update_objective_SingleVector(ids[t], vecs[t].detach().cpu().numpy(), "mean")
    res["mean_cents"] = cf.objective_vectors.clone()
    res["mean_nums"] = cf.objective_vectors_num.clone()
    # nearest label downsample pin (App. A-3)
    res["full_lab"] = full_lab
    res["near_lr"] = F.interpolate(full_lab.reshape(N, 1, 128, 128).float(), size=(h, w), mode="nearest")
    save("meanvec", feat=feat, out=out, lab_lr=lab_lr, cents=cents, **res)


# ------------------------------------------------------------------ G-aspp
def _fill_module(mod, prefix):
    sd = mod.state_dict()
    new = {}
    for k, v in sd.items():
        if v.dim() == 4:
            kind = "conv"
        elif v.dim() == 2:
            kind = "lin"
        elif k.endswith("weight"):
            kind = "gn_w"
        else:
            kind = "bias"
        new[k] = detweights.fill(prefix + k, tuple(v.shape), kind)
    mod.load_state_dict(new)
    return new


def gen_aspp():
    torch.manual_seed(0)
    head = Classifier_Module2(64, [6, 12, 18, 24], [6, 12, 18, 24], 19)
    sd = _fill_module(head, "aspp64.")
    head.eval()
    g = synth.gen(7)
    x = torch.randn((2, 64, 33, 29), generator=g).requires_grad_()
    res = head(x, get_feat=True)
    out, feat = res["out"], res["feat"]
    probe = torch.randn(out.shape, generator=g)
    probe_f = 0.1 * torch.randn(feat.shape, generator=g)
    ((out * probe).sum() + (feat * probe_f).sum()).backward()
    grads = {}
    for k, p in head.named_parameters():
        key = "gw_" + k.replace(".", "_")
        if p.grad.numel() <= 20000:
            grads[key] = p.grad
        else:       # large conv weights: checksum, L1 norm and a strided sample keep the fixture small
            grads[key + "__sum"] = np.array([synth.checksum(p.grad), float(p.grad.abs().sum())])
            grads[key + "__sample"] = p.grad.reshape(-1)[::97].clone()
    save("aspp", x=x, out=out, feat=feat, probe=probe, probe_f=probe_f, gx=x.grad, **grads)


def gen_aspp_dropout():
    """The reference head with nn.Dropout2d(0.1) LIVE (train mode; seg_model_noaux.py:171,207-208): the drawn (image, channel)
    keep mask is read back from the module's own output (`feat` is the dropped tensor: a dropped channel is exactly zero, a kept
    one is GroupNorm output / 0.9) and stored with input, logits, feat and the gradients, so that the oracle's `keep_mask` and
    the HIP path's `chan_scale` can be held to the reference's Dropout2d semantics, not only to each other."""
    torch.manual_seed(0)
    head = Classifier_Module2(64, [6, 12, 18, 24], [6, 12, 18, 24], 19)
    sd = _fill_module(head, "aspp64.")
    head.train()
    g = synth.gen(8)
    x = torch.randn((4, 64, 17, 21), generator=g).requires_grad_()
    torch.manual_seed(123)
    res = head(x, get_feat=True)
    out, feat = res["out"], res["feat"]
    keep = (feat.detach().abs().amax(dim=(2, 3)) > 0).float()              # [4, 256]
    assert 0.02 < float(1 - keep.mean()) < 0.25, float(1 - keep.mean())
    probe = torch.randn(out.shape, generator=g)
    probe_f = 0.1 * torch.randn(feat.shape, generator=g)
    ((out * probe).sum() + (feat * probe_f).sum()).backward()
    grads = {}
    for k, p in head.named_parameters():
        key = "gw_" + k.replace(".", "_")
        if p.grad.numel() <= 20000:
            grads[key] = p.grad
        else:
            grads[key + "__sum"] = np.array([synth.checksum(p.grad), float(p.grad.abs().sum())])
            grads[key + "__sample"] = p.grad.reshape(-1)[::97].clone()
    print("aspp_dropout: dropped", int((1 - keep).sum()), "of", keep.numel(), "channels")
    save("aspp_dropout", x=x, out=out, feat=feat, keep=keep, probe=probe, probe_f=probe_f, gx=x.grad, **grads)


# ------------------------------------------------------------------ G-model
def _ref_model():
    m = SegModel()
    sd = detweights.state_dict(RESNET101)
    missing = set(m.state_dict().keys()) ^ set(sd.keys())
    assert not missing, sorted(missing)[:5]
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    return m


def gen_model():
    m = _ref_model()
    names = list(m.state_dict().keys())
    pnames = [n for n, _ in m.named_parameters()]
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    m.eval()
    g = synth.gen(8)
    x = torch.rand((2, 3, 128, 128), generator=g) * 2 - 1
    with torch.no_grad():
        sh, dp, out, feat = m(x)
    res = dict(x=x, out_eval=out, feat_eval=feat,
               shallow_sum=np.array(synth.checksum(sh)), deep_sum=np.array(synth.checksum(dp)),
               shallow_shape=np.array(sh.shape), deep_shape=np.array(dp.shape))
    # train mode: batch-stat BN, dropout forced off; gradient probe
    m.train()
    m.final.head[0].p = 0.0
    sh, dp, out, feat = m(x)
    probe = torch.randn(out.shape, generator=g)
    (out * probe).sum().backward()
    res.update(out_train=out, feat_train=feat, probe=probe)
    for n in ["layer0.0.weight", "layer1.0.conv1.weight", "layer2.3.conv2.weight",
              "layer3.22.conv3.weight", "layer4.0.downsample.0.weight",
              "final.conv2d_list.3.0.weight", "final.conv2d_list.0.1.weight",
              "final.bottleneck.0.se.0.weight", "final.bottleneck.1.bias", "final.head.1.weight"]:
        p = dict(m.named_parameters())[n]
        res["g_" + n.replace(".", "_")] = np.array([synth.checksum(p.grad), float(p.grad.abs().sum())])
    res["g_head"] = dict(m.named_parameters())["final.head.1.weight"].grad
    res["rm_after"] = m.state_dict()["layer1.0.bn1.running_mean"]
    res["rv_after"] = m.state_dict()["layer4.2.bn3.running_var"]
    res["nbt_after"] = m.state_dict()["layer0.1.num_batches_tracked"]
    # structure pins
    res["n_state"] = np.array(len(names))
    res["n_params"] = np.array(len(pnames))
    res["n_trainable"] = np.array(len(trainable))
    res["numel_params"] = np.array(sum(p.numel() for p in m.parameters()))
    res["numel_trainable"] = np.array(sum(p.numel() for p in m.parameters() if p.requires_grad))
    groups = m.optim_parameters(2.5e-4)
    g1 = list(groups[0]["params"])
    g10 = list(groups[1]["params"])
    ids = {id(p): n for n, p in m.named_parameters()}
    from collections import Counter
    cnt = Counter(ids[id(p)] for p in g1)
    res["g1_entries"] = np.array(len(g1))
    res["g1_unique"] = np.array(len(cnt))
    res["g1_mult_hist"] = np.array([sum(1 for v in cnt.values() if v == k) for k in range(6)])
    res["g10_entries"] = np.array(len(g10))
    res["g1_mult_stem"] = np.array(cnt["layer0.0.weight"])
    res["g1_mult_block"] = np.array(cnt["layer3.5.conv2.weight"])
    res["g1_mult_down"] = np.array(cnt["layer2.0.downsample.0.weight"])
    res["state_keys"] = np.array(names)
    res["param_keys"] = np.array(pnames)
    res["g1_order"] = np.array([ids[id(p)] for p in g1])
    res["g10_order"] = np.array([ids[id(p)] for p in g10])
    save("model", **res)


# ------------------------------------------------------------------ G-fullsize (benchmark geometries, round 3)
FULLSIZE_GRADS = ["layer0.0.weight", "layer1.0.conv1.weight", "layer1.0.downsample.0.weight", "layer2.0.conv2.weight",
                  "layer2.3.conv3.weight", "layer3.0.downsample.0.weight", "layer3.11.conv2.weight", "layer3.22.conv1.weight",
                  "layer4.0.downsample.0.weight", "layer4.0.conv1.weight", "layer4.2.conv2.weight", "layer4.2.conv3.weight",
                  "final.conv2d_list.0.0.weight", "final.conv2d_list.1.0.weight", "final.conv2d_list.2.0.weight",
                  "final.conv2d_list.3.0.weight", "final.conv2d_list.4.0.weight", "final.conv2d_list.1.1.weight",
                  "final.bottleneck.0.se.0.weight", "final.bottleneck.1.weight", "final.bottleneck.1.bias",
                  "final.bottleneck.2.weight"]


def _gen_fullsize(name, H, W, seed, batch=2):
    """Reference SegModel (model_noaux.py:28-46), train mode (batch-statistics BN), Dropout2d off, deterministic
    weights, on TWO crops of the benchmark geometry: 768x768 -> 97x97 map (BASELINE configs[1]) and 512x1024 -> 65x129
    (configs[3]).  The input and the gradient probe are regenerated from their seeds by the test; stored are the
    reference's logits in full, strided samples + checksums of feat and of 22 weight gradients (all four ASPP dilations,
    every stride / downsample transition), the head gradient in full, and BN running statistics after the pass."""
    m = _ref_model()
    m.train()
    m.final.head[0].p = 0.0
    g = synth.gen(seed)
    x = torch.rand((batch, 3, H, W), generator=g) * 2 - 1
    sh, dp, out, feat = m(x)
    probe = torch.randn(out.shape, generator=g)
    (out * probe).sum().backward()
    # (batch > 2 -- the half-benchmark-batch capture: 8 images, 75 272 rows per GEMM --: logits as a strided sample + sums, no eval pass)
    outs = dict(out=out) if batch <= 2 else dict(out_sample=out.detach().reshape(-1)[::7].clone(),
                                                 out_sum=np.array([synth.checksum(out), float(out.detach().abs().sum()), float(out.detach().abs().max())]))
    res = dict(seed=np.array(seed), geometry=np.array([batch, H, W]), **outs, feat_sample=feat.detach().reshape(-1)[::61].clone(),
               feat_sum=np.array([synth.checksum(feat), float(feat.detach().abs().sum())]),
               shallow_sum=np.array([synth.checksum(sh), float(sh.detach().abs().sum())]),
               deep_sum=np.array([synth.checksum(dp), float(dp.detach().abs().sum())]),
               deep_sample=dp.detach().reshape(-1)[::9973].clone())
    named = dict(m.named_parameters())
    for n in FULLSIZE_GRADS:
        gr = named[n].grad
        key = "g_" + n.replace(".", "_")
        res[key + "__sum"] = np.array([synth.checksum(gr), float(gr.abs().sum()), float(gr.norm())])
        step = max(1, gr.numel() // 4096)
        res[key + "__sample"] = gr.reshape(-1)[::step].clone()
        res[key + "__step"] = np.array(step)
    res["g_head"] = named["final.head.1.weight"].grad
    sd = m.state_dict()
    res["rm_layer1"] = sd["layer1.0.bn1.running_mean"]
    res["rv_layer4"] = sd["layer4.2.bn3.running_var"]
    res["rm_layer3"] = sd["layer3.22.bn3.running_mean"]
    if batch <= 2:
        m.eval()
        with torch.no_grad():
            res["out_eval"] = m(x)[2]
    save(name, **res)


def gen_full768b8():
    """Half the benchmark's student batch (8 of 16 images: the reference's 8-image pass peaks at 36 GB resident, 16 images do not fit this
    container's 62 GB): 75 272 rows per pointwise GEMM -- beyond 2^15 and 2^16 rows, four times the tile count of the 2-image capture."""
    _gen_fullsize("full768b8", 768, 768, 8769, batch=8)


def gen_full512x1024b8():
    """The self-training geometry (configs[3]) at the per-GPU batch of its student(cat) pass: 4 + 4 crops of 512x1024 = 8 images,
    67 080 rows per pointwise GEMM."""
    _gen_fullsize("full512x1024b8", 512, 1024, 8513, batch=8)


def gen_full768():
    _gen_fullsize("full768", 768, 768, 8768)


def gen_full512x1024():
    _gen_fullsize("full512x1024", 512, 1024, 8512)


# ------------------------------------------------------------------ G-mit (MiT-B5 encoder, BASELINE configs[4])
def _import_reference_mit():
    """The reference's model/networks/MixTransfomer.py imports timm and mmcv (both absent here).  What it takes from them
    is NOT arithmetic of the forward pass: `DropPath` (stochastic depth: the identity in eval(), which is how the capture
    runs), `to_2tuple` (int -> pair), `trunc_normal_` (initialisation; the capture loads its own deterministic weights),
    `register_model` / `_cfg` (registry decorators, unused), mmcv's `load_checkpoint` / `get_logger` (I/O).  Stand-ins for
    exactly those names let the reference module import; every nn.Module and every tensor op that runs is the reference's."""
    import importlib.util

    class _DropPath(torch.nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            assert not self.training, "the capture runs in eval(): DropPath is the identity"
            return x

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("timm")
    mod("timm.models")
    mod("timm.models.layers", DropPath=_DropPath, to_2tuple=lambda v: (v, v) if not isinstance(v, tuple) else v,
        trunc_normal_=lambda t, std=1.0, **k: t)
    mod("timm.models.registry", register_model=lambda f: f)
    mod("timm.models.vision_transformer", _cfg=lambda **k: k)
    mod("mmcv")
    mod("mmcv.runner", load_checkpoint=lambda *a, **k: None)
    mod("mmcv.utils", get_logger=lambda *a, **k: None)
    spec = importlib.util.spec_from_file_location("ref_mit", os.path.join(REF, "model", "networks", "MixTransfomer.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def gen_mit():
    from oracle import mit as om
    ref = _import_reference_mit()
    torch.manual_seed(0)
    net = ref.mit_b5()
    sd = om.state_dict(om.MIT_B5)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net.eval()                                                  # DropPath off; nothing else depends on the mode
    g = synth.gen(55)
    x = torch.rand((2, 3, 128, 96), generator=g) * 2 - 1        # H != W; stage maps 32x24, 16x12, 8x6, 4x3; 12 keys per stage
    outs = net(x)
    probes = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * p).sum() for o, p in zip(outs, probes)).backward()
    res = {"x": x, "keys": np.array(list(sd.keys()))}
    for i, (o, p) in enumerate(zip(outs, probes)):
        res[f"c{i + 1}"] = o
        res[f"probe{i + 1}"] = p
    named = dict(net.named_parameters())
    res["grad_norms"] = np.array([float(named[k].grad.norm()) for k in sd.keys()])
    res["grad_sums"] = np.array([synth.checksum(named[k].grad) for k in sd.keys()])
    for k in ["patch_embed1.proj.weight", "patch_embed3.proj.weight", "block1.0.attn.sr.weight", "block1.2.attn.q.weight",
              "block2.3.attn.kv.weight", "block3.0.norm1.weight", "block3.20.mlp.fc1.weight", "block3.39.mlp.dwconv.dwconv.weight",
              "block3.39.mlp.fc2.bias", "block4.1.attn.proj.weight", "block4.2.attn.kv.bias", "norm2.weight", "norm4.bias",
              "block2.0.attn.norm.weight", "patch_embed2.norm.bias"]:
        gr = named[k].grad
        step = max(1, gr.numel() // 2048)
        res["g_" + k.replace(".", "_")] = gr.reshape(-1)[::step].clone()
        res["gstep_" + k.replace(".", "_")] = np.array(step)
    save("mit", **res)
    # a second, single-stage-sized capture for the attention geometry of the benchmark: 576 keys (24x24 after sr 8 of a
    # 192x192 map) needs a 768x768 input -- too slow for a CPU capture of B5; mit_b1 on one image is enough to pin indexing
    net1 = ref.mit_b1()
    sd1 = om.state_dict(om.MIT_B1)
    net1.load_state_dict(sd1)
    net1.eval()
    x1 = torch.rand((1, 3, 768, 768), generator=synth.gen(56)) * 2 - 1
    with torch.no_grad():
        o1 = net1(x1)
    # ... and the full-depth mit_b5 (40 blocks in stage 3) on the same image, forward only
    with torch.no_grad():
        o5 = net(x1)
    save("mit768", seed=np.array(56), c1_sample=o1[0].reshape(-1)[::211].clone(), c2_sample=o1[1].reshape(-1)[::53].clone(),
         c3_sample=o1[2].reshape(-1)[::7].clone(), c4_sample=o1[3].reshape(-1)[::3].clone(),
         sums=np.array([float(o.abs().sum()) for o in o1]), maxs=np.array([float(o.abs().max()) for o in o1]),
         b5_c1_sample=o5[0].reshape(-1)[::211].clone(), b5_c2_sample=o5[1].reshape(-1)[::53].clone(),
         b5_c3_sample=o5[2].reshape(-1)[::7].clone(), b5_c4_sample=o5[3].reshape(-1)[::3].clone(),
         b5_sums=np.array([float(o.abs().sum()) for o in o5]), b5_maxs=np.array([float(o.abs().max()) for o in o5]))


def gen_mit768bwd():
    """Backward capture at the BENCHMARK geometry (768x768: 36 864 queries against 576 keys in stage 1, MixTransfomer.py:85-144):
    the reference's mit_b1 on one image, probes on all four stage outputs (regenerated from their seed by the test: c1's probe
    alone is 9 MB), gradient norm of every parameter and strided samples of the gradients whose index arithmetic depends on the
    geometry -- attention q / kv / spatial-reduction conv and the patch embeddings of every stage."""
    from oracle import mit as om
    ref = _import_reference_mit()
    net = ref.mit_b1()
    sd = om.state_dict(om.MIT_B1)
    net.load_state_dict(sd)
    net.eval()
    x = torch.rand((1, 3, 768, 768), generator=synth.gen(56)) * 2 - 1          # the image of mit768.npz
    outs = net(x)
    gp = synth.gen(57)
    probes = [torch.randn(o.shape, generator=gp) for o in outs]
    sum((o * p).sum() for o, p in zip(outs, probes)).backward()
    named = dict(net.named_parameters())
    res = {"seed_x": np.array(56), "seed_probe": np.array(57), "keys": np.array(list(sd.keys())),
           "grad_norms": np.array([float(named[k].grad.norm()) for k in sd.keys()]),
           "out_sums": np.array([float(o.detach().abs().sum()) for o in outs])}
    picks = [k for k in sd.keys() if k.startswith(("patch_embed",)) and k.endswith("proj.weight")]
    picks += [k for k in sd.keys() if ".attn.q.weight" in k or ".attn.kv.weight" in k or ".attn.sr.weight" in k]
    picks += ["block1.0.norm1.weight", "block1.1.mlp.fc1.weight", "block1.1.mlp.dwconv.dwconv.weight", "block2.0.mlp.fc2.weight", "norm1.weight", "norm4.bias"]
    for k in picks:
        gr = named[k].grad
        step = max(1, gr.numel() // 1024)
        res["g_" + k.replace(".", "_")] = gr.reshape(-1)[::step].clone()
        res["gstep_" + k.replace(".", "_")] = np.array(step)
    save("mit768bwd", **res)


# ------------------------------------------------------------------ G-step (warm-up, 3 steps)
# ------------------------------------------------------------------ G-segformer-head (decode head of BASELINE configs[4])
def _import_reference_segformer_head():
    """The reference's model/networks/segformer_head.py imports mmcv (absent).  `normal_init` is initialisation (unused: the
    capture loads deterministic weights); `DepthwiseSeparableConvModule` belongs to DAFormerHead (not captured).  `ConvModule` IS
    arithmetic: SegFormerHead builds `linear_fuse` from it (:63-68).  The stand-in is mmcv 1.x's documented behaviour for that
    call -- Conv2d without bias (bias='auto' with a norm layer), BatchNorm2d, ReLU, children named `conv`, `bn`, `activate` -- so
    that block of the capture is pinned to mmcv's documentation, not its code (oracle/segformer_head.py header says so).  Every other
    module and tensor op that runs (MLP, the resizes, the concatenation order, Dropout2d, linear_pred) is the reference's."""
    import importlib.util

    class _ConvModule(torch.nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None, **kw):
            super().__init__()
            assert norm_cfg is not None and norm_cfg["type"] == "BN" and not kw
            self.conv = torch.nn.Conv2d(in_channels, out_channels, kernel_size, bias=False)
            self.bn = torch.nn.BatchNorm2d(out_channels)
            self.activate = torch.nn.ReLU(inplace=True)

        def forward(self, x):
            return self.activate(self.bn(self.conv(x)))

    mm = types.ModuleType("mmcv")
    cnn = types.ModuleType("mmcv.cnn")
    cnn.ConvModule, cnn.normal_init, cnn.DepthwiseSeparableConvModule = _ConvModule, (lambda *a, **k: None), None
    mm.cnn = cnn
    sys.modules.update({"mmcv": mm, "mmcv.cnn": cnn})
    spec = importlib.util.spec_from_file_location("ref_segformer_head", os.path.join(REF, "model", "networks", "segformer_head.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def gen_segformer_head():
    from oracle import segformer_head as oh
    ref = _import_reference_segformer_head()
    chans = [64, 128, 320, 512]
    head = ref.SegFormerHead(in_channels=chans, channels=128, feature_strides=[4, 8, 16, 32], num_classes=19, in_index=[0, 1, 2, 3],
                             dropout_ratio=0.1, align_corners=False)
    sd = oh.state_dict(chans, 19)
    assert list(head.state_dict().keys()) == list(sd.keys()), (list(head.state_dict().keys()), list(sd.keys()))
    head.load_state_dict(sd)
    head.train()
    head.dropout.p = 0.0                                        # Dropout2d off: its draws are not part of the pin
    g = synth.gen(91)
    # stage maps of a 2 x (3, 128, 96) image: 32x24, 16x12, 8x6, 4x3
    feats = [(torch.randn((2, c, 128 // s, 96 // s), generator=g)).requires_grad_() for c, s in zip(chans, (4, 8, 16, 32))]
    logits, c_raw = head(feats)
    probe = torch.randn(logits.shape, generator=g)
    (logits * probe).sum().backward()
    named = dict(head.named_parameters())
    res = {"keys": np.array(list(sd.keys())), "probe": probe, "logits": logits, "c_raw_sum": np.array(synth.checksum(c_raw)),
           "c_raw_sample": c_raw.detach().reshape(-1)[::397].clone(),
           "running_mean": head.linear_fuse.bn.running_mean, "running_var": head.linear_fuse.bn.running_var,
           "nbt": head.linear_fuse.bn.num_batches_tracked}
    for i, f in enumerate(feats):
        res[f"c{i + 1}"] = f
        res[f"dc{i + 1}"] = f.grad
    for k, p in named.items():
        gr = p.grad
        res["gnorm_" + k.replace(".", "_")] = np.array(float(gr.double().norm()))     # (float64: the fp32 norm of 2.4 M elements is off by 5e-5)
        step = max(1, gr.numel() // 4096)
        res["g_" + k.replace(".", "_")] = gr.reshape(-1)[::step].clone()
        res["gstep_" + k.replace(".", "_")] = np.array(step)
    # eval mode (running statistics as loaded + one update), forward only, a second geometry with odd sizes (100 x 76 image)
    head.eval()
    g2 = synth.gen(92)
    feats2 = [torch.randn((1, c, h, w), generator=g2) for c, (h, w) in zip(chans, ((25, 19), (13, 10), (7, 5), (4, 3)))]
    with torch.no_grad():
        lo2, _ = head(feats2)
    for i, f in enumerate(feats2):
        res[f"e{i + 1}"] = f
    res["logits_eval"] = lo2
    save("segformer_head", **res)


def gen_step():
    import torch.optim as optim
    B, H, W = 2, 128, 128
    student, teacher = _ref_model(), _ref_model()
    for mdl in (student, teacher):
        mdl.final.head[0].p = 0.0
    opt = optim.SGD(student.optim_parameters(2.5e-4), lr=2.5e-4, momentum=0.9, weight_decay=0.0005)
    up = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
    teacher = create_teacher_params(teacher, student)
    random.seed(77)
    log = {"ce": [], "distil": [], "total": [], "lr": []}
    for it in range(3):
        student.train()
        adjust_learning_rate([opt], base_lr=2.5e-4, i_iter=it, max_iter=80000, power=0.9)
        with torch.no_grad():
            teacher = update_teacher_params(teacher, student, it)
        x, x_aug, rec, lab = synth.warmup_batch(1000 + it, B, H, W, block=16)
        mask = torch.zeros(lab.size())
        for i in range(B):
            present = torch.unique(lab[i]).tolist()
            pick = random.sample(present, len(present) // 2)
            if 255 not in pick:
                pick.append(255)
            for c in pick:
                mask[i][lab[i] == c] = 1
        mix = torch.zeros(rec.size())
        for i in range(B):
            mix[i] = torch.mul(rec[i], 1 - mask[i]) + torch.mul(x_aug[i], mask[i])
        cat = torch.cat([x, mix])
        _, _, s_cat, _ = student(cat)
        s_cat = up(s_cat)
        _, _, t_cat, _ = teacher(cat)
        t_cat = up(t_cat)
        ce = cross_entropy2d(s_cat[:B], lab)
        di = distillation_loss(t_cat, s_cat)
        total = 1.0 * ce + 0.5 * di
        opt.zero_grad()
        total.backward()
        opt.step()
        log["ce"].append(float(ce)); log["distil"].append(float(di)); log["total"].append(float(total))
        log["lr"].append(opt.param_groups[0]["lr"])
        print("step", it, log["ce"][-1], log["distil"][-1])
    student.eval()
    teacher.eval()
    xp = synth.warmup_batch(2000, 1, H, W, block=16)[0]
    with torch.no_grad():
        _, _, so, _ = student(xp)
        _, _, to, _ = teacher(xp)
    res = {k: np.array(v, dtype=np.float64) for k, v in log.items()}
    sd = student.state_dict()
    td = teacher.state_dict()
    for n in ["layer0.0.weight", "layer3.10.conv2.weight", "final.head.1.weight",
              "final.conv2d_list.2.0.weight", "layer2.0.downsample.0.weight"]:
        res["ps_" + n.replace(".", "_")] = np.array(synth.checksum(sd[n]))
        res["pt_" + n.replace(".", "_")] = np.array(synth.checksum(td[n]))
    res["student_head"] = sd["final.head.1.weight"]
    res["teacher_head"] = td["final.head.1.weight"]
    res["stu_rm"] = sd["layer1.0.bn1.running_mean"]
    res["tea_rm"] = td["layer1.0.bn1.running_mean"]
    save("step", probe_student=so, probe_teacher=to, **res)


# ------------------------------------------------------------------ G-traj (training trajectories, round 5)
TRAJ_SAMPLES = ["layer0.0.weight", "layer1.0.conv1.weight", "layer2.0.downsample.0.weight", "layer3.0.conv2.weight",
                "layer3.10.conv2.weight", "layer3.22.conv3.weight", "layer4.2.conv2.weight", "final.conv2d_list.2.0.weight",
                "final.bottleneck.1.weight"]


def _gen_traj(name, B, H, W, steps, seed0, block, mix_seed, variant=None):
    """The reference's warm-up loop body (train_DiGA_gta2city_warm_up.py:197-305: adjust_learning_rate, update_teacher_params,
    inline ClassMix, student(cat) / teacher(cat), nn.Upsample(bilinear, align_corners=True), cross_entropy2d + 0.5 *
    distillation_loss, torch.optim.SGD over SegModel.optim_parameters) run for `steps` iterations on the reference's own
    classes, every batch regenerated from its seed (seed0 + it).  Stored per step: CE, distillation loss, total, lr, the student
    head's weight norm and the norm of its change; at the end: the head in full, strided samples + norms of nine trunk / ASPP
    weights (and of their change since step 0), BN running statistics of both networks, eval-mode probe logits of both."""
    import torch.optim as optim
    student, teacher = _ref_model(), _ref_model()
    for mdl in (student, teacher):
        mdl.final.head[0].p = 0.0
    opt = optim.SGD(student.optim_parameters(2.5e-4), lr=2.5e-4, momentum=0.9, weight_decay=0.0005)
    up = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
    teacher = create_teacher_params(teacher, student)
    w0 = {n: student.state_dict()[n].clone() for n in TRAJ_SAMPLES + ["final.head.1.weight"]}
    random.seed(mix_seed)
    log = {"ce": [], "distil": [], "total": [], "lr": [], "head_norm": [], "head_delta": []}
    for it in range(steps):
        student.train()
        adjust_learning_rate([opt], base_lr=2.5e-4, i_iter=it, max_iter=80000, power=0.9)
        with torch.no_grad():
            teacher = update_teacher_params(teacher, student, it)
        x, x_aug, rec, lab = synth.warmup_batch(seed0 + it, B, H, W, block=block)
        mask = torch.zeros(lab.size())
        for i in range(B):
            present = torch.unique(lab[i]).tolist()
            pick = random.sample(present, len(present) // 2)
            if 255 not in pick:
                pick.append(255)
            for c in pick:
                mask[i][lab[i] == c] = 1
        mix = torch.zeros(rec.size())
        for i in range(B):
            mix[i] = torch.mul(rec[i], 1 - mask[i]) + torch.mul(x_aug[i], mask[i])
        cat = torch.cat([x, mix])
        _, _, s_cat, _ = student(cat)
        s_cat = up(s_cat)
        _, _, t_cat, _ = teacher(cat)
        t_cat = up(t_cat)
        ce = cross_entropy2d(s_cat[:B], lab)
        di = distillation_loss(t_cat, s_cat)
        total = 1.0 * ce + 0.5 * di
        opt.zero_grad()
        total.backward()
        opt.step()
        hw = student.state_dict()["final.head.1.weight"]
        log["ce"].append(float(ce)); log["distil"].append(float(di)); log["total"].append(float(total))
        log["lr"].append(opt.param_groups[0]["lr"])
        log["head_norm"].append(float(hw.double().norm())); log["head_delta"].append(float((hw - w0["final.head.1.weight"]).double().norm()))
        print(name, "step", it, log["ce"][-1], log["distil"][-1], log["head_delta"][-1], flush=True)
    student.eval()
    teacher.eval()
    xp = synth.warmup_batch(seed0 + 1000, 1, min(H, 256), min(W, 256), block=block)[0]
    with torch.no_grad():
        _, _, so, _ = student(xp)
        _, _, to, _ = teacher(xp)
    res = {k: np.array(v, dtype=np.float64) for k, v in log.items()}
    res["geometry"] = np.array([B, H, W, steps, seed0, block, mix_seed])
    sd, td = student.state_dict(), teacher.state_dict()
    for n in TRAJ_SAMPLES:
        key = n.replace(".", "_")
        step = max(1, sd[n].numel() // 2048)
        res["ps_" + key + "__sample"] = sd[n].reshape(-1)[::step].clone()
        res["ps_" + key + "__delta_sample"] = (sd[n] - w0[n]).reshape(-1)[::step].clone()
        res["ps_" + key + "__step"] = np.array(step)
        res["ps_" + key + "__norms"] = np.array([float(sd[n].double().norm()), float((sd[n] - w0[n]).double().norm()),
                                                 float(td[n].double().norm()), synth.checksum(sd[n]), synth.checksum(td[n])])
    res["student_head"] = sd["final.head.1.weight"]
    res["teacher_head"] = td["final.head.1.weight"]
    res["student_head_delta"] = sd["final.head.1.weight"] - w0["final.head.1.weight"]
    for n in ["layer1.0.bn1.running_mean", "layer3.22.bn3.running_mean", "layer4.2.bn3.running_var", "layer2.3.bn2.running_var"]:
        res["stu_" + n.replace(".", "_")] = sd[n]
        res["tea_" + n.replace(".", "_")] = td[n]
    res.update(probe_student=so, probe_teacher=to)
    if variant is not None:
        return res
    # The trajectory's own sensitivity to fp32 ROUNDING: the same loop, the same reference classes, run a second time with
    # torch's CPU convolutions on their other implementation (oneDNN switched off: im2col + GEMM instead of oneDNN's blocked
    # direct kernels -- same mathematics, another summation order).  What separates the two reference runs after `steps` steps is
    # the floor any other fp32 implementation is measured against (tests/test_gpu_trajectory.py): `floor_*`.
    with torch.backends.mkldnn.flags(enabled=False):
        alt = _gen_traj(name + "/no-onednn", B, H, W, steps, seed0, block, mix_seed, variant="no_onednn")

    def rel(a, b):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        return float((a - b).norm() / b.norm().clamp_min(1e-300))

    res["floor_ce_dev"] = np.abs(alt["ce"] - res["ce"]) / np.abs(res["ce"])
    res["floor_distil_dev"] = np.abs(alt["distil"] - res["distil"]) / np.abs(res["distil"])
    res["floor_head_delta"] = np.array(rel(alt["student_head_delta"], res["student_head_delta"]))
    res["floor_trunk_delta"] = np.array([rel(alt[k], res[k]) for k in sorted(res) if k.endswith("__delta_sample")])
    res["floor_probe"] = np.array([float((alt[k] - res[k]).abs().max() / res[k].abs().max()) for k in ("probe_student", "probe_teacher")])
    res["floor_bn"] = np.array([float((alt[k] - res[k]).abs().max() / res[k].abs().max()) for k in sorted(res)
                                if k.startswith(("stu_layer", "tea_layer"))])
    print(name, "rounding floor: loss", float(res["floor_ce_dev"].max()), float(res["floor_distil_dev"].max()), "head change",
          float(res["floor_head_delta"]), "trunk change", float(res["floor_trunk_delta"].max()), "probe", res["floor_probe"].tolist(),
          "bn", float(res["floor_bn"].max()), flush=True)
    save(name, **res)


def gen_traj25():
    _gen_traj("traj25", 2, 128, 128, 25, 5000, 16, 79)


def gen_traj768():
    _gen_traj("traj768", 1, 768, 768, 3, 6000, 32, 80)


# ------------------------------------------------------------------ G-selftrain (one self-training step)
def gen_selftrain():
    import torch.optim as optim
    B, H, W = 2, 128, 128
    student, teacher = _ref_model(), _ref_model()
    for mdl in (student, teacher):
        mdl.final.head[0].p = 0.0
    opt = optim.SGD(student.optim_parameters(2.5e-4), lr=2.5e-4, momentum=0.9, weight_decay=0.0005)
    up = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
    teacher = create_teacher_params(teacher, student)
    cf = Class_Features(numbers=19)
    cents0 = torch.randn((19, 256), generator=synth.gen(7)) * 0.3
    cf.objective_vectors = cents0.clone()
    random.seed(78)
    it = 3
    student.train()
    adjust_learning_rate([opt], base_lr=2.5e-4, i_iter=it, max_iter=80000, power=0.9)
    with torch.no_grad():
        teacher = update_teacher_params(teacher, student, it)
    x, x_aug, rec, lab, t_img, t_aug, pseudo_prob = synth.selftrain_batch(3000, B, H, W, block=16)
    # --- self_training.py:259-275 re-enacted (ClassMix #1)
    mask = torch.zeros(lab.size())
    for i in range(B):
        present = torch.unique(lab[i]).tolist()
        pick = random.sample(present, len(present) // 2)
        if 255 not in pick:
            pick.append(255)
        for c in pick:
            mask[i][lab[i] == c] = 1
    mix = torch.zeros(rec.size())
    for i in range(B):
        mix[i] = torch.mul(rec[i], 1 - mask[i]) + torch.mul(x_aug[i], mask[i])
    cat = torch.cat([x, mix])
    _, _, s_cat, _ = student(cat)
    with torch.no_grad():
        _, _, t_cat_lr, t_feat_cat = teacher(cat)
    t_aug_raw = t_cat_lr[B:]
    t_cat = up(t_cat_lr)
    s_feat_tea_aug = t_feat_cat[B:]
    # --- :298-304 bilateral consensus
    with torch.no_grad():
        pseudo = pseudo_prob.clone()
        _, _, tt_pred, tt_feat = teacher(t_img)
        fw = up(cf.get_centroid_weight(tt_feat.detach()))
        feat_pseudo = fw.max(1, keepdim=True)[1].squeeze(1)
        pseudo[pseudo_prob != feat_pseudo] = 255
    # --- :306-325 ClassMix #2 with label paste
    cross_lab = pseudo.clone()
    mask = torch.zeros(lab.size())
    for i in range(B):
        present = torch.unique(lab[i]).tolist()
        pick = random.sample(present, len(present) // 2)
        if 255 not in pick:
            pick.append(255)
        for c in pick:
            cross_lab[i][lab[i] == c] = c
            mask[i][lab[i] == c] = 1
    cross_mix = torch.zeros(t_aug.size())
    for i in range(B):
        cross_mix[i] = torch.mul(t_aug[i], 1 - mask[i]) + torch.mul(x[i], mask[i])
    cross_lab = cross_lab.long()
    # --- :327-341 centroid updates (target, then source)
    with torch.no_grad():
        nl_t = F.interpolate(pseudo.clone().reshape([B, 1, H, W]).float(), size=tt_feat.size()[2:], mode="nearest")
        v_t, id_t = cf.calculate_mean_vector(tt_feat, tt_pred.detach(), nl_t)
        for k in range(len(id_t)):
            cf.update_objective_SingleVector(id_t[k], v_t[k].detach(), start_mean=False)
        nl_s = F.interpolate(lab.clone().reshape([B, 1, H, W]).float(), size=s_feat_tea_aug.size()[2:], mode="nearest")
        v_s, id_s = cf.calculate_mean_vector(s_feat_tea_aug, t_aug_raw.detach(), nl_s)
        for k in range(len(id_s)):
            cf.update_objective_SingleVector(id_s[k], v_s[k].detach(), start_mean=False)
    _, _, c_pred, _ = student(cross_mix)
    c_pred = up(c_pred)
    s_up = up(s_cat)
    ce = cross_entropy2d(s_up[:B], lab)
    di = distillation_loss(t_cat, s_up)
    ce_mix = cross_entropy2d(c_pred, cross_lab)
    total = 1.0 * (ce + ce_mix) + 0.25 * di
    opt.zero_grad()
    total.backward()
    opt.step()
    sd = student.state_dict()
    top2 = fw.topk(2, dim=1)[0]
    save("selftrain", ce=ce, distil=di, ce_mix=ce_mix, total=total, cents0=cents0, cents=cf.objective_vectors,
         nums=cf.objective_vectors_num, ids_t=np.array(id_t), ids_s=np.array(id_s), pseudo=pseudo,
         feat_pseudo=feat_pseudo, margin=(top2[:, 0] - top2[:, 1]), cross_lab=cross_lab,
         student_head=sd["final.head.1.weight"], kept=np.array(float((pseudo != 255).float().mean())),
         ps_layer3=np.array(synth.checksum(sd["layer3.10.conv2.weight"])))
    print("selftrain", float(ce), float(di), float(ce_mix), "kept", float((pseudo != 255).float().mean()),
          "ids_t", id_t, "ids_s", id_s)


# ------------------------------------------------------------------ G-selftraj (self-training trajectory, round 5)
def _gen_selftraj(name, B, H, W, steps, seed0, block, mix_seed, variant=None):
    """The reference's self-training loop body (train_DiGA_gta2city_self_training.py:214-387: learning-rate schedule, EMA teacher,
    ClassMix #1, student(cat) / teacher(cat), bilateral consensus of the offline pseudo-labels with the centroid pseudo-labeler,
    ClassMix #2 with label paste, the two centroid-bank updates, student(cross_mix), CE + CE_mix + 0.25 * distillation, SGD) run for
    `steps` iterations on the reference's own classes (SegModel, Class_Features, cross_entropy2d, distillation_loss, ...), every batch
    regenerated from its seed (seed0 + it).  Per step: the three losses, the share of pseudo-labels the consensus keeps, the
    norm of the centroid bank and of its change; at the end: the bank and its counts, the student head and its change, BatchNorm
    running statistics, eval-mode probe logits.  `floor_*`: the same loop a second time with oneDNN off (see _gen_traj)."""
    import torch.optim as optim
    student, teacher = _ref_model(), _ref_model()
    for mdl in (student, teacher):
        mdl.final.head[0].p = 0.0
    opt = optim.SGD(student.optim_parameters(2.5e-4), lr=2.5e-4, momentum=0.9, weight_decay=0.0005)
    up = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
    teacher = create_teacher_params(teacher, student)
    cf = Class_Features(numbers=19)
    cents0 = torch.randn((19, 256), generator=synth.gen(7)) * 0.3
    cf.objective_vectors = cents0.clone()
    w0 = student.state_dict()["final.head.1.weight"].clone()
    random.seed(mix_seed)
    log = {k: [] for k in ("ce", "distil", "ce_mix", "total", "kept", "cents_norm", "cents_delta", "head_delta")}
    ids_log = []

    def pick_classes(lab_i):
        present = torch.unique(lab_i).tolist()
        pick = random.sample(present, len(present) // 2)
        if 255 not in pick:
            pick.append(255)
        return pick

    for it in range(steps):
        student.train()
        adjust_learning_rate([opt], base_lr=2.5e-4, i_iter=it, max_iter=80000, power=0.9)
        with torch.no_grad():
            teacher = update_teacher_params(teacher, student, it)
        x, x_aug, rec, lab, t_img, t_aug, pseudo_prob = synth.selftrain_batch(seed0 + it, B, H, W, block=block)
        mask = torch.zeros(lab.size())                                   # :259-275 ClassMix #1
        for i in range(B):
            for c in pick_classes(lab[i]):
                mask[i][lab[i] == c] = 1
        mix = torch.zeros(rec.size())
        for i in range(B):
            mix[i] = torch.mul(rec[i], 1 - mask[i]) + torch.mul(x_aug[i], mask[i])
        cat = torch.cat([x, mix])
        _, _, s_cat, _ = student(cat)
        with torch.no_grad():
            _, _, t_cat_lr, t_feat_cat = teacher(cat)
        t_aug_raw = t_cat_lr[B:]
        t_cat = up(t_cat_lr)
        s_feat_tea_aug = t_feat_cat[B:]
        with torch.no_grad():                                            # :298-304 bilateral consensus
            pseudo = pseudo_prob.clone()
            _, _, tt_pred, tt_feat = teacher(t_img)
            fw = up(cf.get_centroid_weight(tt_feat.detach()))
            feat_pseudo = fw.max(1, keepdim=True)[1].squeeze(1)
            pseudo[pseudo_prob != feat_pseudo] = 255
        cross_lab = pseudo.clone()                                       # :306-325 ClassMix #2 with label paste
        mask = torch.zeros(lab.size())
        for i in range(B):
            for c in pick_classes(lab[i]):
                cross_lab[i][lab[i] == c] = c
                mask[i][lab[i] == c] = 1
        cross_mix = torch.zeros(t_aug.size())
        for i in range(B):
            cross_mix[i] = torch.mul(t_aug[i], 1 - mask[i]) + torch.mul(x[i], mask[i])
        cross_lab = cross_lab.long()
        with torch.no_grad():                                            # :327-341 centroid updates (target, then source)
            nl_t = F.interpolate(pseudo.clone().reshape([B, 1, H, W]).float(), size=tt_feat.size()[2:], mode="nearest")
            v_t, id_t = cf.calculate_mean_vector(tt_feat, tt_pred.detach(), nl_t)
            for k in range(len(id_t)):
                cf.update_objective_SingleVector(id_t[k], v_t[k].detach(), start_mean=False)
            nl_s = F.interpolate(lab.clone().reshape([B, 1, H, W]).float(), size=s_feat_tea_aug.size()[2:], mode="nearest")
            v_s, id_s = cf.calculate_mean_vector(s_feat_tea_aug, t_aug_raw.detach(), nl_s)
            for k in range(len(id_s)):
                cf.update_objective_SingleVector(id_s[k], v_s[k].detach(), start_mean=False)
        _, _, c_pred, _ = student(cross_mix)
        c_pred = up(c_pred)
        s_up = up(s_cat)
        ce = cross_entropy2d(s_up[:B], lab)
        di = distillation_loss(t_cat, s_up)
        ce_mix = cross_entropy2d(c_pred, cross_lab)
        total = 1.0 * (ce + ce_mix) + 0.25 * di
        opt.zero_grad()
        total.backward()
        opt.step()
        hw = student.state_dict()["final.head.1.weight"]
        cents = torch.as_tensor(cf.objective_vectors)
        for k, v in (("ce", ce), ("distil", di), ("ce_mix", ce_mix), ("total", total), ("kept", (pseudo != 255).float().mean()),
                     ("cents_norm", cents.double().norm()), ("cents_delta", (cents - cents0).double().norm()),
                     ("head_delta", (hw - w0).double().norm())):
            log[k].append(float(v))
        ids_log.append([len(id_t), len(id_s)])
        print(name, "step", it, log["ce"][-1], log["distil"][-1], log["ce_mix"][-1], "kept", log["kept"][-1], "cents", log["cents_delta"][-1],
              flush=True)
    student.eval()
    teacher.eval()
    xp = synth.warmup_batch(seed0 + 1000, 1, H, W, block=block)[0]
    with torch.no_grad():
        _, _, so, _ = student(xp)
        _, _, to, _ = teacher(xp)
    res = {k: np.array(v, dtype=np.float64) for k, v in log.items()}
    res["geometry"] = np.array([B, H, W, steps, seed0, block, mix_seed])
    res["n_ids"] = np.array(ids_log)
    sd, td = student.state_dict(), teacher.state_dict()
    res.update(cents0=cents0, cents=torch.as_tensor(cf.objective_vectors).clone(), nums=torch.as_tensor(cf.objective_vectors_num).clone().float(),
               student_head=sd["final.head.1.weight"], student_head_delta=sd["final.head.1.weight"] - w0,
               teacher_head=td["final.head.1.weight"], probe_student=so, probe_teacher=to)
    for n in ["layer1.0.bn1.running_mean", "layer3.22.bn3.running_mean", "layer4.2.bn3.running_var", "layer2.3.bn2.running_var"]:
        res["stu_" + n.replace(".", "_")] = sd[n]
        res["tea_" + n.replace(".", "_")] = td[n]
    if variant is not None:
        return res
    # TWO further runs of the reference that differ from the capture in summation order only: (A) oneDNN off (im2col + GEMM instead of
    # blocked direct kernels), (B) round 6: one thread instead of eight (other partial-sum trees in the weight-gradient and BatchNorm
    # reductions).  The floor is the LARGER of the two deviations per quantity: this loop takes discrete decisions every step (which
    # pseudo-labels survive the consensus), so one pair of runs is one draw of a heavy-tailed quantity -- round 6's HIP path touched
    # 3.03 x the one-pair CE floor at one step after its loss block's reduction order changed (bound: 3 x).
    with torch.backends.mkldnn.flags(enabled=False):
        alt_a = _gen_selftraj(name + "/no-onednn", B, H, W, steps, seed0, block, mix_seed, variant="no_onednn")
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        alt_b = _gen_selftraj(name + "/one-thread", B, H, W, steps, seed0, block, mix_seed, variant="one_thread")
    finally:
        torch.set_num_threads(threads)

    def rel(a, b):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        return float((a - b).norm() / b.norm().clamp_min(1e-300))

    def floors(alt):
        f = {}
        for k in ("ce", "distil", "ce_mix"):
            f["floor_" + k + "_dev"] = np.abs(alt[k] - res[k]) / np.abs(res[k])
        f["floor_kept"] = np.abs(alt["kept"] - res["kept"])
        f["floor_cents"] = np.array(rel(alt["cents"] - alt["cents0"], res["cents"] - res["cents0"]))
        f["floor_head_delta"] = np.array(rel(alt["student_head_delta"], res["student_head_delta"]))
        f["floor_probe"] = np.array([float((alt[k] - res[k]).abs().max() / res[k].abs().max()) for k in ("probe_student", "probe_teacher")])
        f["floor_bn"] = np.array([float((alt[k] - res[k]).abs().max() / res[k].abs().max()) for k in sorted(res)
                                  if k.startswith(("stu_layer", "tea_layer"))])
        f["floor_nums_equal"] = np.array(bool(torch.equal(torch.as_tensor(alt["nums"]), torch.as_tensor(res["nums"]))))
        return f

    fa, fb = floors(alt_a), floors(alt_b)
    for k in fa:
        res[k] = np.logical_and(fa[k], fb[k]) if k == "floor_nums_equal" else np.maximum(fa[k], fb[k])
        res[k.replace("floor_", "floorA_")] = fa[k]
        res[k.replace("floor_", "floorB_")] = fb[k]
    print(name, "rounding floor: losses", [float(res["floor_" + k + "_dev"].max()) for k in ("ce", "distil", "ce_mix")], "kept",
          float(res["floor_kept"].max()), "centroid change", float(res["floor_cents"]), "head change", float(res["floor_head_delta"]),
          "probe", res["floor_probe"].tolist(), "bn", float(res["floor_bn"].max()), "counts equal", bool(res["floor_nums_equal"]), flush=True)
    save(name, **res)


def gen_selftraj10():
    _gen_selftraj("selftraj10", 2, 128, 128, 10, 8000, 16, 81)


# ------------------------------------------------------------------ translator (SURVEY 8f "next" #1)
def gen_translator():
    from model.model_noaux import ImgDecoder, ImgEncoder  # reference
    torch.manual_seed(0)
    enc, dec = ImgEncoder(), ImgDecoder()
    keys = {}
    for tag, m in (("enc", enc), ("dec", dec)):
        sd = m.state_dict()
        new = {k: detweights.fill(f"translator.{tag}.{k}", tuple(v.shape), "conv" if v.dim() == 4 else "bias")
               for k, v in sd.items()}
        m.load_state_dict(new)
        m.eval()
        keys[tag] = list(sd.keys())
    g = synth.gen(12)
    x = torch.rand((2, 3, 64, 96), generator=g) * 2 - 1
    with torch.no_grad():
        feat = enc(x)
        rec = dec(feat)
    save("translator", x=x, feat_sum=np.array(synth.checksum(feat)), feat_shape=np.array(feat.shape), rec=rec,
         enc_keys=np.array(keys["enc"]), dec_keys=np.array(keys["dec"]))


# ------------------------------------------------------------------ G-miou
def gen_miou():
    g = synth.gen(9)
    gt = torch.randint(0, 19, (2, 64, 64), generator=g)
    gt[torch.rand((2, 64, 64), generator=g) < 0.1] = 255
    pred = torch.where(torch.rand((2, 64, 64), generator=g) < 0.6, gt.clamp(max=18),
                       torch.randint(0, 19, (2, 64, 64), generator=g))
    rs = runningScore(19)
    import contextlib
    import io
    rs.update(gt.numpy(), pred.numpy())
    with contextlib.redirect_stdout(io.StringIO()):
        sc, cls_iu = rs.get_scores()
    save("miou", gt=gt, pred=pred, hist=rs.confusion_matrix, miou=np.array(sc["Mean IoU : \t"]),
         acc=np.array(sc["Overall Acc: \t"]), acc_cls=np.array(sc["Mean Acc : \t"]),
         fwavacc=np.array(sc["FreqW Acc : \t"]), iu=np.array([cls_iu[i] for i in range(19)]))


# ------------------------------------------------------------------ G-valmiou (two-scale validation mIoU)
def gen_valmiou():
    """The reference's validation pass (G5/evaluate_val.py:73-93 = warm_up.py:346-359) re-enacted on a seeded synthetic
    val set with the deterministic ResNet-101 weights: eval() model on the image and on its half-size bilinear
    (align_corners) downscale, both logit maps upsampled to label size, element-wise max, argmax, runningScore.
    Ground truth = the reference's own prediction with 30 % of 16x16 blocks re-drawn (so the score is neither 0 nor 1
    and every pixel whose argmax flips moves it).  Inputs are regenerated from the seed by the test; stored: labels,
    prediction, confusion matrix, scores."""
    import contextlib
    import io
    m = _ref_model().eval()
    n_img, H, W = 3, 256, 512
    up = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
    rs = runningScore(19)
    preds, gts, margins = [], [], []
    for i in range(n_img):
        g = synth.gen(7000 + i)
        img = torch.rand((1, 3, H, W), generator=g) * 2.0 - 1.0 + 0.5 * torch.randn((1, 3, 1, 1), generator=g)
        img_ds = F.interpolate(img, (H // 2, W // 2), mode="bilinear", align_corners=True)
        with torch.no_grad():
            pred = up(m(img)[2])
            pred_ds = up(m(img_ds)[2])
        fused = torch.max(pred, pred_ds)
        top2 = fused.topk(2, dim=1).values
        margins.append((top2[:, 0] - top2[:, 1])[0])
        p = fused.max(1)[1]
        noisy = synth.block_labels(g, 1, H, W, 16)
        flip = (torch.rand((1, H // 16, W // 16), generator=g) < 0.3).repeat_interleave(16, 1).repeat_interleave(16, 2)
        gt = torch.where(flip, noisy, p)
        rs.update(gt.numpy(), p.numpy())
        preds.append(p[0])
        gts.append(gt[0])
    with contextlib.redirect_stdout(io.StringIO()):
        sc, cls_iu = rs.get_scores()
    save("valmiou", gt=torch.stack(gts).to(torch.uint8), pred=torch.stack(preds).to(torch.uint8),
         near_ties=np.array([int((torch.stack(margins) < t).sum()) for t in (1e-4, 1e-3, 1e-2)]), hist=rs.confusion_matrix, miou=np.array(sc["Mean IoU : \t"]),
         acc=np.array(sc["Overall Acc: \t"]), iu=np.array([cls_iu[i] for i in range(19)]),
         geometry=np.array([n_img, H, W]), seed0=np.array(7000))


def gen_valmiou_full():
    """The reference's validation pass at ITS geometry (G5/evaluate_val.py:60,73-93; warm_up.py:346-359): ONE 1024 x 2048 image
    and its 512 x 1024 bilinear (align_corners) half through SegModel.eval() (running-statistics BatchNorm, N = 1: 129 x 257 and
    65 x 129 maps = 33 153 / 8 385 rows per GEMM), both logit maps upsampled to 1024 x 2048, element-wise max, argmax,
    runningScore.  Stored: the low-resolution logits of both scales in full (19 x 129 x 257 and 19 x 65 x 129 floats), the
    prediction (uint8), a packed mask of the pixels whose top-2 margin of the fused logits is below 1e-3 of the logit scale (where a
    rounding-level difference may flip the argmax), ground truth (the prediction with 30 % of 64 x 64 blocks re-drawn), the 19 x 19
    confusion matrix and the scores.  The image is regenerated from its seed by the test."""
    import contextlib
    import io
    m = _ref_model().eval()
    H, W = 1024, 2048
    up = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
    g = synth.gen(7100)
    img = torch.rand((1, 3, H, W), generator=g) * 2.0 - 1.0 + 0.5 * torch.randn((1, 3, 1, 1), generator=g)
    img_ds = F.interpolate(img, (H // 2, W // 2), mode="bilinear", align_corners=True)
    with torch.no_grad():
        lo = m(img)[2]
        lo_ds = m(img_ds)[2]
        fused = torch.max(up(lo), up(lo_ds))
    top2 = fused.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])[0]
    scale = float(fused.abs().max())
    p = fused.max(1)[1]
    noisy = synth.block_labels(g, 1, H, W, 64)
    flip = (torch.rand((1, H // 64, W // 64), generator=g) < 0.3).repeat_interleave(64, 1).repeat_interleave(64, 2)
    gt = torch.where(flip, noisy, p)
    rs = runningScore(19)
    rs.update(gt.numpy(), p.numpy())
    with contextlib.redirect_stdout(io.StringIO()):
        sc, cls_iu = rs.get_scores()
    near = (margin < 1e-3 * scale).numpy()
    print("valmiou_full: logits", tuple(lo.shape), tuple(lo_ds.shape), "scale", scale, "classes predicted", torch.unique(p).tolist(),
          "near ties", int(near.sum()), "mIoU", sc["Mean IoU : \t"], flush=True)
    save("valmiou_full", logits=lo[0], logits_ds=lo_ds[0], pred=p[0].to(torch.uint8), gt=gt[0].to(torch.uint8),
         near_tie_bits=np.packbits(near), near_ties=np.array([int((margin < t * scale).sum()) for t in (1e-5, 1e-4, 1e-3)]),
         logit_scale=np.array(scale), hist=rs.confusion_matrix, miou=np.array(sc["Mean IoU : \t"]), acc=np.array(sc["Overall Acc: \t"]),
         iu=np.array([cls_iu[i] for i in range(19)]), geometry=np.array([H, W]), seed=np.array(7100))


# ------------------------------------------------------------------ G-trainmiou (round 6: a TRAINED model's validation mIoU)
TRAINMIOU = dict(B=2, H=128, W=128, steps=300, block=32, every=50, n_val=32, lr=2.5e-4, seeds=(0, 1, 2),
                 data_seed0=20000, val_seed0=90000)


def _two_scale_val(student, cfg):
    """G5/evaluate_val.py:73-93 = warm_up.py:343-360 on `n_val` held-out images of the learnable task: eval() student on the
    image and on its half-size bilinear (align_corners) downscale, both upsampled to label size, element-wise max, argmax,
    runningScore."""
    import contextlib
    import io
    H, W = cfg["H"], cfg["W"]
    up = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
    rs = runningScore(19)
    student.eval()
    for i in range(cfg["n_val"]):
        img, _, _, gt = synth.learnable_batch(cfg["val_seed0"] + i, 1, H, W, block=cfg["block"])
        img_ds = F.interpolate(img, (H // 2, W // 2), mode="bilinear", align_corners=True)
        with torch.no_grad():
            pred = up(student(img)[2])
            pred_ds = up(student(img_ds)[2])
        p = torch.max(pred, pred_ds).max(1)[1]
        rs.update(gt.numpy(), p.numpy())
    with contextlib.redirect_stdout(io.StringIO()):
        sc, cls_iu = rs.get_scores()
    student.train()
    return float(sc["Mean IoU : \t"]), float(sc["Overall Acc: \t"]), np.array([cls_iu[i] for i in range(19)]), rs.confusion_matrix.copy()


def _train_and_validate(cfg, seed):
    """The reference's warm-up loop (train_DiGA_gta2city_warm_up.py:197-305) with Dropout2d LIVE (torch's global RNG seeded with
    `seed`), ClassMix drawn from Python's `random` seeded with `seed`, batches of the learnable task from data_seed0 +
    1000 * seed + it; the reference's validation (:343-373) every `every` steps and at the end."""
    import torch.optim as optim
    B, H, W = cfg["B"], cfg["H"], cfg["W"]
    torch.manual_seed(1234 + seed)
    random.seed(4321 + seed)
    student, teacher = _ref_model(), _ref_model()
    opt = optim.SGD(student.optim_parameters(cfg["lr"]), lr=cfg["lr"], momentum=0.9, weight_decay=0.0005)
    up = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
    teacher = create_teacher_params(teacher, student)
    curve, acc_curve, ce_log, di_log = [], [], [], []
    for it in range(cfg["steps"]):
        student.train()
        adjust_learning_rate([opt], base_lr=cfg["lr"], i_iter=it, max_iter=80000, power=0.9)
        with torch.no_grad():
            teacher = update_teacher_params(teacher, student, it)
        x, x_aug, rec, lab = synth.learnable_batch(cfg["data_seed0"] + 1000 * seed + it, B, H, W, block=cfg["block"])
        mask = torch.zeros(lab.size())
        for i in range(B):
            present = torch.unique(lab[i]).tolist()
            pick = random.sample(present, len(present) // 2)
            if 255 not in pick:
                pick.append(255)
            for c in pick:
                mask[i][lab[i] == c] = 1
        mix = torch.zeros(rec.size())
        for i in range(B):
            mix[i] = torch.mul(rec[i], 1 - mask[i]) + torch.mul(x_aug[i], mask[i])
        cat = torch.cat([x, mix])
        _, _, s_cat, _ = student(cat)
        s_cat = up(s_cat)
        _, _, t_cat, _ = teacher(cat)
        t_cat = up(t_cat)
        ce = cross_entropy2d(s_cat[:B], lab)
        di = distillation_loss(t_cat, s_cat)
        total = 1.0 * ce + 0.5 * di
        opt.zero_grad()
        total.backward()
        opt.step()
        ce_log.append(float(ce)); di_log.append(float(di))
        if (it + 1) % cfg["every"] == 0:
            miou, acc, _, _ = _two_scale_val(student, cfg)
            curve.append(miou); acc_curve.append(acc)
            print(f"trainmiou seed {seed} step {it + 1}: CE {np.mean(ce_log[-cfg['every']:]):.4f} distil {np.mean(di_log[-cfg['every']:]):.4f} "
                  f"val mIoU {100 * miou:.2f} acc {100 * acc:.2f}", flush=True)
    miou, acc, iu, hist = _two_scale_val(student, cfg)
    return dict(curve=np.array(curve), acc_curve=np.array(acc_curve), ce=np.array(ce_log), distil=np.array(di_log),
                miou=miou, acc=acc, iu=iu, hist=hist)


def gen_trainmiou():
    """north_star's training-level criterion ('mIoU within 0.1 of reference on fixed seed'): after a few optimizer steps two fp32
    implementations are no longer pointwise comparable (traj25: probe logits 9e-3 of scale apart after 25 steps, reference
    against reference), so the criterion is checked the only way it can be -- statistically.  The reference's warm-up loop runs
    `steps` iterations on a LEARNABLE synthetic task (oracle/synth.py::learnable_batch), Dropout2d live, for every seed of
    `seeds`; the reference's two-scale validation gives the mIoU curve and the final mIoU on held-out images.  Stored: per-seed
    curves, final mIoU / accuracy / per-class IoU / confusion matrix, loss logs, the mean and seed-to-seed spread of the final
    mIoU -- what tests/test_gpu_trainmiou.py holds the HIP path's own runs against."""
    cfg = dict(TRAINMIOU)
    for k in ("steps", "every", "n_val", "block"):
        if os.environ.get("TRAINMIOU_" + k.upper()):
            cfg[k] = int(os.environ["TRAINMIOU_" + k.upper()])
    if os.environ.get("TRAINMIOU_LR"):
        cfg["lr"] = float(os.environ["TRAINMIOU_LR"])
    if os.environ.get("TRAINMIOU_SEEDS"):
        cfg["seeds"] = tuple(int(v) for v in os.environ["TRAINMIOU_SEEDS"].split(","))
    runs = [_train_and_validate(cfg, s) for s in cfg["seeds"]]
    final = np.array([r["miou"] for r in runs])
    print("trainmiou final mIoU per seed", (100 * final).tolist(), "mean", 100 * final.mean(), "spread (max-min)", 100 * (final.max() - final.min()),
          "std", 100 * final.std(ddof=1) if len(final) > 1 else 0.0, flush=True)
    if os.environ.get("TRAINMIOU_DRY"):
        return
    save("trainmiou", seeds=np.array(cfg["seeds"]), curve=np.stack([r["curve"] for r in runs]), acc_curve=np.stack([r["acc_curve"] for r in runs]),
         ce=np.stack([r["ce"] for r in runs]), distil=np.stack([r["distil"] for r in runs]), miou=final,
         acc=np.array([r["acc"] for r in runs]), iu=np.stack([r["iu"] for r in runs]), hist=np.stack([r["hist"] for r in runs]),
         geometry=np.array([cfg["B"], cfg["H"], cfg["W"], cfg["steps"], cfg["block"], cfg["every"], cfg["n_val"], cfg["data_seed0"], cfg["val_seed0"]]),
         lr=np.array(cfg["lr"]))


# ------------------------------------------------------------------ G-ohem ("next" row 4)
def gen_ohem():
    """OhemCrossEntropy (G5/util/loss.py:65-122) in its three regimes: threshold = thresh (many uncertain
    pixels), threshold = k-th smallest probability (confident predictions, min_kept larger than the number of
    uncertain pixels), and low-res scores that the module upsamples itself."""
    g = synth.gen(11)
    out = {}
    # (a) near-uniform predictions: the k-th smallest p_t is far below 0.7 -> threshold 0.7
    xa = torch.randn((2, 19, 33, 33), generator=g).requires_grad_()
    ya = striped_labels(g, (2, 33, 33))
    la = OhemCrossEntropy(min_kept=50)(xa, ya)
    la.backward()
    out.update(a_x=xa, a_y=ya, a_loss=la, a_grad=xa.grad, a_min_kept=np.array(50))
    # (b) confident predictions: most p_t > 0.7, so sorted[min_kept] > 0.7 becomes the threshold
    yb = striped_labels(g, (2, 33, 33))
    xb = torch.randn((2, 19, 33, 33), generator=g)
    onehot = F.one_hot(yb.clamp(max=18), 19).permute(0, 3, 1, 2).float()
    xb = (xb + 6.0 * onehot * (torch.rand((2, 1, 33, 33), generator=g) < 0.9)).requires_grad_()
    lb = OhemCrossEntropy(min_kept=600)(xb, yb)
    lb.backward()
    out.update(b_x=xb, b_y=yb, b_loss=lb, b_grad=xb.grad, b_min_kept=np.array(600))
    # (c) min_kept beyond the number of valid pixels: k = n_valid - 1 (the largest probability), everything below it kept
    lc_x = xb.detach().clone().requires_grad_()
    lc = OhemCrossEntropy(min_kept=100000)(lc_x, yb)
    lc.backward()
    out.update(c_loss=lc, c_grad=lc_x.grad)
    # (d) low-res scores: the module upsamples (bilinear, align_corners) to the label size first
    xd = (2.0 * torch.randn((2, 19, 9, 9), generator=g)).requires_grad_()
    yd = striped_labels(g, (2, 65, 65))
    ld = OhemCrossEntropy(min_kept=300, thres=0.5)(xd, yd)
    ld.backward()
    out.update(d_x=xd, d_y=yd, d_loss=ld, d_grad=xd.grad, d_min_kept=np.array(300), d_thres=np.array(0.5))
    save("ohem", **out)


ALL = dict(trainmiou=gen_trainmiou, aspp_dropout=gen_aspp_dropout, valmiou_full=gen_valmiou_full, full768b8=gen_full768b8, full512x1024b8=gen_full512x1024b8, traj25=gen_traj25, traj768=gen_traj768, selftraj10=gen_selftraj10, mit=gen_mit, segformer_head=gen_segformer_head, mit768bwd=gen_mit768bwd, full768=gen_full768, full512x1024=gen_full512x1024, ohem=gen_ohem, ce=gen_ce, distill=gen_distill, upsample=gen_upsample, ema=gen_ema, sgd=gen_sgd,
           classmix=gen_classmix, centroid=gen_centroid, meanvec=gen_meanvec, aspp=gen_aspp,
           model=gen_model, step=gen_step, selftrain=gen_selftrain, translator=gen_translator, miou=gen_miou, valmiou=gen_valmiou)

if __name__ == "__main__":
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(ALL)
    for n in names:
        print("==", n)
        ALL[n]()
