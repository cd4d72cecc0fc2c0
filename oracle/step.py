"""Oracle (CPU, test infrastructure only): one whole training step.

Restates the loop bodies of
  * G5/train_DiGA_gta2city_warm_up.py:197-305         (warm-up)
  * G5/train_DiGA_gta2city_self_training.py:214-387   (self-training)
on top of the other oracle modules, with torch autograd on CPU.  Visualisation,
logging, data loading, the kornia augmentation and the frozen translator are
inputs/outside the path.  Pinned by tests/golden/step.npz.
"""
import random

import torch

from . import centroids as oc
from . import classmix as ocm
from . import deeplab as od
from . import losses as ol
from . import optim as oo

_FROZEN = ("bn_w", "bn_b")
_BUFFERS = ("bn_rm", "bn_rv", "bn_nbt")


def param_keys(arch=od.RESNET101):
    """Keys of nn.Module.parameters() in reference order (buffers excluded)."""
    return [k for k, (_, kind) in od.state_shapes(arch).items() if kind not in _BUFFERS]


def trainable_keys(arch=od.RESNET101):
    return [k for k, (_, kind) in od.state_shapes(arch).items()
            if kind not in _BUFFERS and kind not in _FROZEN]


def multiplicity(key):
    """How often a trainable tensor occurs in the optimizer groups that
    G5/model/model_noaux.py:48-77 builds."""
    if key.startswith("final."):
        return 1                                   # 10x group: plain parameters()
    if key.startswith("layer0."):
        return 2                                   # Sequential + the conv itself
    if ".downsample." in key:
        return 4                                   # layer, block, downsample Sequential, conv
    return 3                                       # layer, block, conv


def lr_scale(key):
    return 10.0 if key.startswith("final.") else 1.0


class Trainer:
    """Student + EMA teacher + duplicate-aware SGD, all on state dicts."""

    def __init__(self, student_sd, teacher_sd, arch=od.RESNET101, base_lr=2.5e-4, max_iter=80000,
                 power=0.9, momentum=0.9, weight_decay=5e-4, droprate_off=True, keep_masks=None):
        """droprate_off=False keeps the head's Dropout2d(arch.droprate) LIVE (seg_model_noaux.py:171,207-208: student and teacher both
        run it in train mode); its draws then come from `keep_masks(role, n, width)` -> a 0/1 [n, width] tensor, role in
        ("student", "teacher"), called once per forward pass in the order the passes are listed in the step."""
        self.arch = arch if not droprate_off else od.Arch(**{**arch.__dict__, "droprate": 0.0})
        if not droprate_off and keep_masks is None:
            raise ValueError("oracle.step.Trainer: live dropout needs the drawn keep masks (keep_masks=...)")
        self.keep_masks = None if droprate_off else keep_masks
        self.s, self.t = student_sd, teacher_sd
        self.pkeys, self.tkeys = param_keys(arch), trainable_keys(arch)
        self.bufs = {k: torch.zeros_like(self.s[k]) for k in self.tkeys}
        self.base_lr, self.max_iter, self.power = base_lr, max_iter, power
        self.momentum, self.wd = momentum, weight_decay
        self.first = True
        with torch.no_grad():                      # create_teacher_params: copy parameters only
            for k in self.pkeys:
                self.t[k].copy_(self.s[k])

    def _keep(self, n, width=None, role="student"):
        width = self.arch.aspp_width if width is None else width
        if self.keep_masks is not None:
            return self.keep_masks(role, n, width)
        return torch.ones(n, width)                # dropout forced off (mask of ones, p=0)

    def _sgd(self, grads, it):
        lr = oo.poly_lr(self.base_lr, it, self.max_iter, self.power)
        ks = self.tkeys
        oo.sgd_step_dup([self.s[k] for k in ks], [grads[k] for k in ks], [self.bufs[k] for k in ks],
                        [multiplicity(k) for k in ks], [lr * lr_scale(k) for k in ks],
                        self.momentum, self.wd, first_step=self.first)
        self.first = False
        return lr

    def warmup_step(self, it, x, x_aug, rec, labels, rng=random, lambda_seg=1.0, lambda_distil=0.5):
        B = x.shape[0]
        oo.ema_update([self.t[k] for k in self.pkeys], [self.s[k] for k in self.pkeys], it)
        mix, _, _ = ocm.classmix(rec, x_aug, labels, rng)
        cat = torch.cat([x, mix])
        leaves = {k: self.s[k].detach().requires_grad_() for k in self.tkeys}
        sd = {**self.s, **leaves}
        _, _, s_lr, _ = od.forward(sd, cat, self.arch, training=True, keep_mask=self._keep(2 * B),
                                   update_stats=True)
        with torch.no_grad():
            _, _, t_lr, _ = od.forward(self.t, cat, self.arch, training=True,
                                       keep_mask=self._keep(2 * B, role="teacher"), update_stats=True)
        total, ce, di = ol.warmup_losses_lowres(s_lr, t_lr, labels, lambda_seg, lambda_distil)
        grads = dict(zip(leaves.keys(), torch.autograd.grad(total, list(leaves.values()))))
        lr = self._sgd(grads, it)
        return {"ce": float(ce), "distil": float(di), "total": float(total), "lr": lr}

    def selftrain_step(self, it, x, x_aug, rec, labels, t_img, t_aug, pseudo_prob, centroids, nums, rng=random,
                       lambda_seg=1.0, lambda_distil=0.25, momentum=1e-4):
        """self_training.py:214-387.  centroids [19,256] / nums [19] are updated in place."""
        B = x.shape[0]
        oo.ema_update([self.t[k] for k in self.pkeys], [self.s[k] for k in self.pkeys], it)
        mix, _, _ = ocm.classmix(rec, x_aug, labels, rng)                       # :259-275
        cat = torch.cat([x, mix])
        leaves = {k: self.s[k].detach().requires_grad_() for k in self.tkeys}
        sd = {**self.s, **leaves}
        _, _, s_lr, _ = od.forward(sd, cat, self.arch, training=True, keep_mask=self._keep(2 * B),
                                   update_stats=True)                            # :281
        with torch.no_grad():
            _, _, t_lr, t_feat = od.forward(self.t, cat, self.arch, training=True, keep_mask=self._keep(2 * B, role="teacher"),
                                            update_stats=True)                   # :286-287
            _, _, tt_lr, tt_feat = od.forward(self.t, t_img, self.arch, training=True, keep_mask=self._keep(B, role="teacher"),
                                              update_stats=True)                 # :300
            w = oc.centroid_weight(tt_feat, centroids)                           # :301
            pseudo, _ = oc.consensus_filter(w, pseudo_prob)                      # :302-304
            cross_mix, _, _, cross_lab = ocm.classmix(t_aug, x, labels, rng, bg_labels=pseudo)   # :306-325
            hw = tt_feat.shape[-2:]
            for feat, out, lab in ((tt_feat, tt_lr, pseudo), (t_feat[B:], t_lr[B:], labels)):    # :327-341
                lab_lr = oc.nearest_downsample_labels(lab, hw)
                vecs, ids, _ = oc.class_mean_vectors(feat, out, lab_lr)
                oc.centroid_ema_apply(centroids, nums, vecs, ids, momentum)
        _, _, c_lr, _ = od.forward(sd, cross_mix, self.arch, training=True, keep_mask=self._keep(B),
                                   update_stats=True)                            # :343-344
        total_s, ce, di = ol.warmup_losses_lowres(s_lr, t_lr, labels, lambda_seg, lambda_distil)
        ce_mix = ol.cross_entropy2d(ol.upsample_bilinear_ac(c_lr, cross_lab.shape[-2:]), cross_lab)
        total = total_s + lambda_seg * ce_mix                                    # :348-356,382
        grads = dict(zip(leaves.keys(), torch.autograd.grad(total, list(leaves.values()))))
        lr = self._sgd(grads, it)
        return {"ce": float(ce), "distil": float(di), "ce_mix": float(ce_mix), "total": float(total), "lr": lr,
                "kept": float((pseudo != 255).float().mean())}
