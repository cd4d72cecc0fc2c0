"""The SegFormer-B5 / MiT student of BASELINE.json configs[4]: "SegFormer-B5 (MiT transformer) backbone variant, fp16 --
MFMA attention path for the distillation student".

The reference ships the encoder (G5/model/networks/MixTransfomer.py) and the SegFormer decode head
(G5/model/networks/segformer_head.py:25-165) but wires them into nothing (networks/__init__.py:8-9 exports only HRNet / OCRNet;
SURVEY section 2.1), so THE WIRING BELOW IS THE BUILD'S OWN: encoder + head take the place of the ResNet-101 trunk and ASPP
classifier inside the reference's `SegModel` interface (model_noaux.py:28-46,48-77) -- same 4-tuple output, same two optimizer
groups (encoder at 1x, head at 10x the learning rate).

head="segformer" (default): the SegFormer all-MLP head on all four stages, logits at 1/4 scale -- a SegFormer-B5 as published
(diga_amd/model/networks/segformer_head.py; evaluated with the fuse conv folded into the per-stage embeddings).
head="aspp": the reference's DeepLab classifier `Classifier_Module2` (seg_model_noaux.py:140-214) on the last stage only (512
channels, stride 32, logits 24x24 for a 768x768 crop) -- the round-3 wiring, a much lighter workload than a SegFormer; kept for
comparison (bench.py --c5-head aspp).

Everything downstream (upsample + CE + distillation block, EMA teacher, ClassMix, fused SGD, gradient all-reduce) is the
DeepLab path's code unchanged, so `DigaTrainer` drives it as it drives `SegModel`.
Encoder: fp16 storage / fp32 accumulate (diga_amd/model/networks/MixTransfomer.py); head: fp32 tensors, the conv arithmetic
selected by _lib.set_conv_math like every other DigaConv2d."""
import os
import sys

import torch.nn as nn

_pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib  # noqa: E402
from diga_amd.model.networks import MixTransfomer as mit  # noqa: E402
from diga_amd.model.networks.segformer_head import SegFormerHead  # noqa: E402
from diga_amd.model.seg_model_noaux import Classifier_Module2  # noqa: E402


class SegFormerStudent(nn.Module):
    def __init__(self, backbone="mit_b5", n_classes=19, drop_path_rate=None, head="segformer"):
        super().__init__()
        if head not in ("segformer", "aspp"):
            raise ValueError("SegFormerStudent: head is 'segformer' or 'aspp'")
        self.n_classes = n_classes
        self.head_kind = head
        self.backbone = getattr(mit, backbone)()
        if drop_path_rate is not None:
            self.backbone.reset_drop_path(drop_path_rate)
        if head == "aspp":
            self.final = Classifier_Module2(self.backbone.embed_dims[-1], [6, 12, 18, 24], [6, 12, 18, 24], n_classes)
        else:
            self.final = SegFormerHead(in_channels=list(self.backbone.embed_dims), channels=128, feature_strides=[4, 8, 16, 32],
                                       num_classes=n_classes, in_index=[0, 1, 2, 3], dropout_ratio=0.1, align_corners=False)
            self.final.init_weights()

    def forward(self, x):
        """x [N,3,H,W] -> (c2 [N,128,H/8,W/8], c4 [N,512,H/32,W/32], logits, feat): logits [N,19,H/4,W/4] and the fused feature
        [N,768,H/4,W/4] with the SegFormer head; [N,19,H/32,W/32] and [N,256,H/32,W/32] with the ASPP head."""
        _lib.require_gpu(x)
        c1, c2, c3, c4 = self.backbone(x)
        if self.head_kind == "aspp":
            res = self.final(c4)
            return c2, c4, res['out'], res['feat']
        logits, feat = self.final([c1, c2, c3, c4])
        return c2, c4, logits, feat

    def set_head_dropout(self, p):
        """Dropout2d rate of whichever head is wired (the parity tests switch it off: its draws are not part of any pin)."""
        drop = self.final.head[0] if self.head_kind == "aspp" else self.final.dropout
        if drop is not None:
            drop.p = float(p)

    @property
    def grad_overflow(self):
        """Device flag of the fp16 backward's overflow check (diga_amd/model/networks/MixTransfomer.py): DigaSGD.step(found_inf=...)."""
        return self.backbone.grad_overflow

    def adjust_loss_scale(self, *a, **k):
        return self.backbone.adjust_loss_scale(*a, **k)

    @property
    def loss_scale(self):
        return self.backbone.loss_scale

    def get_1x_lr_params_NOscale(self):
        for p in self.backbone.parameters():
            if p.requires_grad:
                yield p

    def get_10x_lr_params(self):
        yield from self.final.parameters()

    def optim_parameters(self, learning_rate):
        return [{'params': self.get_1x_lr_params_NOscale(), 'lr': 1 * learning_rate},
                {'params': self.get_10x_lr_params(), 'lr': 10 * learning_rate}]
