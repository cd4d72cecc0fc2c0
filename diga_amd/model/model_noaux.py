"""SegModel with the reference's interface (G5/model/model_noaux.py:10-77):
    forward(x[N,3,H,W]) -> (layer2 output, layer4 output, logits[N,19,h,w], feat[N,256,h,w])
    optim_parameters(lr) -> [{1x group (with the reference's duplicate entries)}, {10x group}]
state_dict keys: layer0.{0,1}.*, layer{1..4}.<i>.*, final.*  (SURVEY section 5.4).
"""
import os
import sys

import torch
import torch.nn as nn

_pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib
from diga_amd.model import norm as dn  # noqa: E402,F401  (fails loudly when the HIP library is missing)
from diga_amd.model.seg_model_noaux import RESNET101, DeeplabMulti  # noqa: E402
from diga_amd.model.translator import ImgDecoder, ImgEncoder  # noqa: E402,F401  (same import surface as the reference)

pspnet_specs = {'n_classes': 19, 'input_size': (713, 713), 'block_config': [3, 4, 23, 3]}


class SegModel(nn.Module):
    def __init__(self, initialization=None, bn_clr=False, arch=RESNET101):
        super().__init__()
        self.n_classes = arch.n_classes
        self.bn_clr = bn_clr
        self.initialization = initialization
        self.arch = arch
        net = DeeplabMulti(pretrained=True, num_classes=self.n_classes, initialization=initialization,
                           bn_clr=bn_clr, arch=arch)
        self.layer0 = nn.Sequential(net.conv1, net.bn1, net.relu, net.maxpool)
        self.layer1, self.layer2, self.layer3, self.layer4 = net.layer1, net.layer2, net.layer3, net.layer4
        if bn_clr:
            self.bn_pretrain = net.bn_pretrain
        self.final = net.layer5

    def forward(self, x):
        _lib.require_gpu(x)
        if self.training:
            dn.bump_batches_tracked(self)                     # one launch for the 104 BatchNorm step counters
        conv1, bn1, _relu, maxpool = self.layer0              # ReLU is fused into the BN kernel
        shallow = self.layer2(self.layer1(maxpool(bn1(conv1(x), relu=True))))
        deep = self.layer4(self.layer3(shallow))
        if self.bn_clr:
            deep = self.bn_pretrain(deep)
        res = self.final(deep)
        return shallow, deep, res['out'], res['feat']

    # ---- optimizer groups -------------------------------------------------------------------
    def get_1x_lr_params_NOscale(self):
        """Backbone parameters as the reference enumerates them: for every sub-module of layer0..layer4
        (the containers included) all its trainable parameters -- so a conv weight is yielded once per
        enclosing container (2x stem, 3x block convs, 4x downsample convs)."""
        for stage in (self.layer0, self.layer1, self.layer2, self.layer3, self.layer4):
            for module in stage.modules():
                for p in module.parameters():
                    if p.requires_grad:
                        yield p

    def get_10x_lr_params(self):
        if self.bn_clr:
            yield from self.bn_pretrain.parameters()
        yield from self.final.parameters()

    def optim_parameters(self, learning_rate):
        return [{'params': self.get_1x_lr_params_NOscale(), 'lr': 1 * learning_rate},
                {'params': self.get_10x_lr_params(), 'lr': 10 * learning_rate}]
