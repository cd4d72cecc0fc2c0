"""DeepLabV2 / ResNet-101 (output stride 8) with the ProDA-style ASPP head on the MI355X kernels --
module tree and state_dict keys of the reference (G5/model/seg_model_noaux.py:57-101 Bottleneck, :122-137
SEBlock, :140-214 Classifier_Module2, :216-261 ResNetMulti), so checkpoints are interchangeable.

Every layer is a drop-in subclass of the torch module the reference uses, computing with the HIP kernels
behind include/diga_hip.h: DigaConv2d (fp32-MFMA implicit GEMM), DigaBatchNorm2d (batch statistics in train
mode, frozen affine; fused +ReLU / +residual), DigaGroupNorm (fused +ReLU / Dropout2d scale, writes into the
concat buffer), SE pool/gate kernels, DigaMaxPool3x3s2.  Only the two tiny SE linears (1280->80->1280 on an
[N,1280] matrix) are library GEMMs.

Semantics kept from the reference: BN uses batch statistics whenever the module is in
train mode (the teacher always is) and updates its running statistics; GroupNorm(32) is trainable; `feat` is
the post-Dropout2d tensor; all conv weights start as N(0, 0.01).
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn

from diga_amd import config
from diga_amd.model import norm as dn
from diga_amd.model.conv import DigaConv2d, takes_twin_only_input


@dataclass(frozen=True)
class Arch:
    layers: Tuple[int, ...] = (3, 4, 23, 3)
    planes: Tuple[int, ...] = (64, 128, 256, 512)
    strides: Tuple[int, ...] = (1, 2, 1, 1)
    dilations: Tuple[int, ...] = (1, 1, 2, 4)
    stem: int = 64
    expansion: int = 4
    aspp_dilations: Tuple[int, ...] = (6, 12, 18, 24)
    aspp_width: int = 256
    gn_groups: int = 32
    se_reduction: int = 16
    n_classes: int = 19
    droprate: float = 0.1


RESNET101 = Arch()
TINY = Arch(layers=(1, 1, 2, 1), planes=(16, 32, 64, 128), stem=16)


def _frozen_bn(channels):
    bn = dn.DigaBatchNorm2d(channels, affine=True)
    for p in bn.parameters():
        p.requires_grad = False
    return bn


class Bottleneck(nn.Module):
    """1x1 (carries the stride) -> 3x3 (carries the dilation) -> 1x1, each followed by frozen-affine BN;
    ReLU and the residual add are fused into the BN kernels."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = DigaConv2d(inplanes, planes, 1, stride=stride, bias=False)
        self.bn1 = _frozen_bn(planes)
        self.conv2 = DigaConv2d(planes, planes, 3, stride=1, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = _frozen_bn(planes)
        self.conv3 = DigaConv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = _frozen_bn(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)          # kept for module-tree parity; the ReLUs run inside the BN kernels
        self.downsample = downsample
        self.stride = stride
        for conv in (self.conv1, self.conv2, self.conv3) + ((downsample[0],) if downsample is not None else ()):
            conv.emit_bn_stats = True              # BN statistics come out of the conv epilogue

    def forward(self, x):
        # bn1's output is read only by conv2 (and conv2's weight gradient): when those run on the split-twin kernels
        # the BN writes the twin instead of the fp32 tensor (no separate conversion pass)
        # ... and the dx of bn2's backward is read only by conv2's backward-data and backward-weight: a twin as well
        # Under autograd the same holds for conv3 (bn2's output / bn3's dx): its weight gradient is 35-39 % faster on twins.
        grad = torch.is_grad_enabled()
        chain = None
        if self.downsample is not None:
            # two convolutions read x here (conv1 and the downsample conv).  Where both are eligible (stride 1) their
            # input gradients are summed through the backward-data epilogues like the ASPP branches' (model/conv.py,
            # `chain`), and the one that runs last also finishes the gradient of the BatchNorm that produced x; else
            # they meet in an autograd add and that BatchNorm keeps its own mask / reduce passes
            box = getattr(x, "_diga_bn_box", None)
            if (grad and x.requires_grad and dn.fuse_backward_enabled() and tuple(self.conv1.stride) == (1, 1)
                    and tuple(self.downsample[0].stride) == (1, 1) and config.active().junction_chain):
                chain = {"remaining": 2, "acc": None, "box": box}
            elif box is not None:
                x._diga_bn_box = None
        tw2 = takes_twin_only_input(self.conv2)
        # (round 2: with streaming epilogue stores the twin kernel is also the faster one for pointwise layers without a
        #  weight gradient -- the no-grad teacher takes it too; config.twin_conv3 = "2" restricts it to autograd passes as in round 1)
        c3 = config.active().twin_conv3
        tw3 = (grad or c3 != "2") and c3 != "0" and takes_twin_only_input(self.conv3, pointwise_ok=True)
        y1 = self.conv1(x, chain=chain)
        # (two forward fusions were built, measured slower and retired in round 6 -- tools/experiments/: bn1's apply pass inside conv2's
        #  Winograd input transform, +5 ms per C2 step; bn3's residual junction inside the next block's conv1 GEMM, +5 ms)
        y = self.bn1(y1, relu=True, twin_out=tw2)
        y = self.bn2(self.conv2(y, twin_grad=tw2 and grad), relu=True, twin_out=tw3, dx_twin=tw2 and grad)
        if self.downsample is None:
            skip = x
        elif chain is not None:
            skip = self.downsample[1](self.downsample[0](x, chain=chain))
        else:
            skip = self.downsample(x)
        return self.bn3(self.conv3(y, twin_grad=tw3), residual=skip, relu=True, dx_twin=tw3)


class SEBlock(nn.Module):
    def __init__(self, inplanes, r=16):
        super().__init__()
        self.global_pool = nn.AdaptiveAvgPool2d((1, 1))     # module-tree parity; pooling runs in diga_avgpool_nhwc
        self.se = nn.Sequential(nn.Linear(inplanes, inplanes // r), nn.ReLU(inplace=True),
                                nn.Linear(inplanes // r, inplanes), nn.Sigmoid())

    def forward(self, x):
        # (the Sequential is the parameter container with the reference's state-dict keys se.0.* / se.2.*; its Linear -> ReLU -> Linear
        #  -> Sigmoid runs on two launches of diga_small_linear_fwd instead of two GEMM-library calls and two elementwise kernels)
        hidden = dn.small_linear(dn.global_avg_pool(x), self.se[0], 1)
        for hook in self.se[1]._forward_hooks.values():         # (the ReLU module is fused away: its forward hooks still see its output)
            hook(self.se[1], (hidden,), hidden)
        return dn.channel_gate(x, dn.small_linear(hidden, self.se[2], 2))


class Classifier_Module2(nn.Module):
    """ASPP head: 1x1 + dilated 3x3 branches (conv+bias -> GN -> ReLU) written side by side into one NHWC
    buffer (no torch.cat), SE gate, 3x3 conv, GN (+ Dropout2d scale) -> feat, 1x1 -> logits."""

    def __init__(self, inplanes, dilation_series, padding_series, num_classes, droprate=0.1, use_se=True,
                 width=256, groups=32, se_reduction=16):
        super().__init__()
        self.width = width

        def branch(k, d, p):
            return nn.Sequential(DigaConv2d(inplanes, width, k, stride=1, padding=p, dilation=d, bias=True),
                                 dn.DigaGroupNorm(groups, width), nn.ReLU(inplace=True))

        self.conv2d_list = nn.ModuleList([branch(1, 1, 0)] +
                                         [branch(3, d, p) for d, p in zip(dilation_series, padding_series)])
        for seq in self.conv2d_list:
            seq[0].share_twin = True               # the five branches read the same tensor: one split twin serves all
        cat = width * (len(dilation_series) + 1)
        tail = [DigaConv2d(cat, width, 3, stride=1, padding=1, bias=True), dn.DigaGroupNorm(groups, width)]
        self.bottleneck = nn.Sequential(*([SEBlock(cat, se_reduction)] if use_se else []), *tail)
        self.head = nn.Sequential(nn.Dropout2d(droprate), DigaConv2d(width, num_classes, 1, bias=False))
        # Effective init of the reference (seg_model_noaux.py:173-198 + the global loop :236-242): only the
        # 3x3 bottleneck conv gets a zero bias; branch biases and the SE linears keep torch's defaults
        # (the reference's isinstance() tests see Sequential/SEBlock containers, not the layers inside).
        nn.init.zeros_(self.bottleneck[-2].bias)

    def _drop_scale(self, n, c, device):
        drop = self.head[0]
        if not drop.training or drop.p <= 0.0:
            return None
        keep = (torch.rand((n, c), device=device) >= drop.p).to(torch.float32)
        return keep / (1.0 - drop.p)

    def forward(self, x, get_feat=True):
        n, _, h, w = x.shape
        nb = len(self.conv2d_list)
        buf = torch.empty((n, h, w, self.width * nb), dtype=torch.float32, device=x.device)
        slices = []
        chain = None
        if torch.is_grad_enabled() and x.requires_grad and dn.fuse_backward_enabled():
            # the branches' input gradients are summed through their backward-data epilogues (model/conv.py)
            chain = {"remaining": nb, "acc": None, "box": getattr(x, "_diga_bn_box", None)}
        for b, (conv, gn, _) in enumerate(self.conv2d_list):
            dst = dn.alias_slice(buf, b * self.width, (b + 1) * self.width)
            slices.append(gn(conv(x, chain=chain), relu=True, out=dst))
        y = dn.assemble(buf, self.width, slices)
        mods = list(self.bottleneck)
        if len(mods) == 3:
            y = mods[0](y)
        y = mods[-2](y)
        feat = mods[-1](y, chan_scale=self._drop_scale(n, self.width, x.device))
        out = self.head[1](feat)
        return {'feat': feat, 'out': out} if get_feat else out


class ResNetMulti(nn.Module):
    def __init__(self, block, layers, num_classes, bn_clr=False, arch=RESNET101):
        super().__init__()
        self.bn_clr = bn_clr
        self.inplanes = arch.stem
        self.conv1 = DigaConv2d(3, arch.stem, 7, stride=2, padding=3, bias=False)
        self.bn1 = _frozen_bn(arch.stem)
        self.conv1.emit_bn_stats = True
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = dn.DigaMaxPool3x3s2()
        stages = [self._make_layer(block, arch.planes[i], layers[i], arch.strides[i], arch.dilations[i])
                  for i in range(4)]
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        self.layer5 = Classifier_Module2(self.inplanes, list(arch.aspp_dilations), list(arch.aspp_dilations),
                                         num_classes, arch.droprate, True, arch.aspp_width, arch.gn_groups,
                                         arch.se_reduction)
        if bn_clr:
            self.bn_pretrain = dn.DigaBatchNorm2d(self.inplanes, affine=True)
            for p in self.bn_pretrain.parameters():
                p.requires_grad = False
        for m in self.modules():                     # the reference's global init runs after the head's own
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.01)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion or dilation in (2, 4):
            down = nn.Sequential(DigaConv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                 _frozen_bn(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, dilation=dilation, downsample=down)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def stem(self, x):
        return self.maxpool(self.bn1(self.conv1(x), relu=True))

    def forward(self, x):
        if self.training:
            dn.bump_batches_tracked(self)
        x = self.layer4(self.layer3(self.layer2(self.layer1(self.stem(x)))))
        if self.bn_clr:
            x = self.bn_pretrain(x)
        return self.layer5(x)


def DeeplabMulti(pretrained=True, num_classes=19, initialization=None, bn_clr=False, arch=RESNET101):
    """Builds the network.  The reference downloads ImageNet/COCO weights here
    (seg_model_noaux.py:7,328); there is no network on the target machines, so weights come from
    `initialization` (a checkpoint path with a 'state_dict' entry) or stay at their N(0,0.01) init
    until load_state_dict() is called."""
    model = ResNetMulti(Bottleneck, list(arch.layers), num_classes, bn_clr=bn_clr, arch=arch)
    if pretrained and initialization is not None:
        saved = torch.load(initialization, map_location="cpu")
        saved = saved.get('state_dict', saved)
        own = model.state_dict()
        own.update({k: v for k, v in saved.items() if k in own and v.shape == own[k].shape})
        model.load_state_dict(own)
    return model
