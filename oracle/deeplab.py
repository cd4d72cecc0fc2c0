"""Oracle (CPU, test infrastructure only): DeepLabV2 / ResNet forward as a pure
function of a state dict.

Restates
  * G5/model/model_noaux.py:10-46          SegModel (layer0..layer4, final)
  * G5/model/seg_model_noaux.py:57-101     Bottleneck (conv1 carries the stride,
                                           conv2 the dilation; BN affine frozen)
  * G5/model/seg_model_noaux.py:122-137    SEBlock
  * G5/model/seg_model_noaux.py:140-214    Classifier_Module2 (ASPP head)
  * G5/model/seg_model_noaux.py:216-261    ResNetMulti (OS-8: layer3 dil 2, layer4 dil 4,
                                           maxpool ceil_mode)
State-dict key names are the reference's.  Differentiable
through torch autograd; BN runs on batch statistics when training=True
 and Dropout2d takes an injected keep mask so that runs
are RNG-free.  Pinned by tests/golden (G-aspp, G-model).
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class Arch:
    layers: Tuple[int, ...] = (3, 4, 23, 3)          # ResNet-101
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
# Small variant for CPU-sized parity cases (BASELINE config 1 stand-in; SURVEY section 7 hard part 6).
TINY = Arch(layers=(1, 1, 2, 1), planes=(16, 32, 64, 128), stem=16)


def _bn(sd, pfx, x, training, momentum=0.1, eps=1e-5, update_stats=False):
    rm, rv = sd[pfx + ".running_mean"], sd[pfx + ".running_var"]
    if training and not update_stats:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, sd[pfx + ".weight"], sd[pfx + ".bias"],
                        training, momentum, eps)


def _bottleneck(sd, pfx, x, stride, dilation, has_down, training, update_stats):
    out = F.conv2d(x, sd[pfx + ".conv1.weight"], stride=stride)
    out = F.relu(_bn(sd, pfx + ".bn1", out, training, update_stats=update_stats))
    out = F.conv2d(out, sd[pfx + ".conv2.weight"], padding=dilation, dilation=dilation)
    out = F.relu(_bn(sd, pfx + ".bn2", out, training, update_stats=update_stats))
    out = F.conv2d(out, sd[pfx + ".conv3.weight"])
    out = _bn(sd, pfx + ".bn3", out, training, update_stats=update_stats)
    if has_down:
        res = F.conv2d(x, sd[pfx + ".downsample.0.weight"], stride=stride)
        res = _bn(sd, pfx + ".downsample.1", res, training, update_stats=update_stats)
    else:
        res = x
    return F.relu(out + res)


def bottleneck_fixed_masks(sd, pfx, x, stride, dilation, has_down, masks):
    """_bottleneck in train mode with its three ReLUs replaced by multiplications with given 0/1 masks
    (masks = activation patterns observed on the device under test).  The block's gradients are discontinuous
    where a pre-activation crosses zero; with the switches pinned the function is smooth around the operating
    point, so gradients can be compared elementwise with a tight bound.  Where a mask disagrees with the sign of
    this function's own pre-activation that value is within rounding of zero, so the forward value is unchanged
    to the same precision.  Same reference lines as _bottleneck (seg_model_noaux.py:81-101)."""
    m1, m2, m3 = masks
    out = F.conv2d(x, sd[pfx + ".conv1.weight"], stride=stride)
    out = _bn(sd, pfx + ".bn1", out, True) * m1
    out = F.conv2d(out, sd[pfx + ".conv2.weight"], padding=dilation, dilation=dilation)
    out = _bn(sd, pfx + ".bn2", out, True) * m2
    out = F.conv2d(out, sd[pfx + ".conv3.weight"])
    out = _bn(sd, pfx + ".bn3", out, True)
    if has_down:
        res = F.conv2d(x, sd[pfx + ".downsample.0.weight"], stride=stride)
        res = _bn(sd, pfx + ".downsample.1", res, True)
    else:
        res = x
    return (out + res) * m3


def trunk(sd, x, arch=RESNET101, training=False, update_stats=False):
    """layer0..layer4; returns (layer2 output, layer4 output)."""
    x = F.conv2d(x, sd["layer0.0.weight"], stride=2, padding=3)
    x = F.relu(_bn(sd, "layer0.1", x, training, update_stats=update_stats))
    x = F.max_pool2d(x, 3, 2, 1, ceil_mode=True)
    shallow = None
    inplanes = arch.stem
    for li in range(4):
        planes, stride, dil = arch.planes[li], arch.strides[li], arch.dilations[li]
        for bi in range(arch.layers[li]):
            first = bi == 0
            has_down = first and (stride != 1 or inplanes != planes * arch.expansion
                                  or dil in (2, 4))
            x = _bottleneck(sd, f"layer{li + 1}.{bi}", x, stride if first else 1, dil,
                            has_down, training, update_stats)
            inplanes = planes * arch.expansion
        if li == 1:
            shallow = x
    return shallow, x


def aspp_head(sd, x, arch=RESNET101, keep_mask=None, pfx="final"):
    """Classifier_Module2.forward(get_feat=True).  keep_mask: None (eval: dropout
    off) or [N,256] of 0/1 (train: Dropout2d with that keep pattern)."""
    g = arch.gn_groups
    branches = []
    for b in range(1 + len(arch.aspp_dilations)):
        w, bias = sd[f"{pfx}.conv2d_list.{b}.0.weight"], sd[f"{pfx}.conv2d_list.{b}.0.bias"]
        if b == 0:
            y = F.conv2d(x, w, bias)
        else:
            d = arch.aspp_dilations[b - 1]
            y = F.conv2d(x, w, bias, padding=d, dilation=d)
        y = F.group_norm(y, g, sd[f"{pfx}.conv2d_list.{b}.1.weight"],
                         sd[f"{pfx}.conv2d_list.{b}.1.bias"], 1e-5)
        branches.append(F.relu(y))
    cat = torch.cat(branches, 1)
    pooled = cat.mean(dim=(2, 3))
    z = F.relu(F.linear(pooled, sd[f"{pfx}.bottleneck.0.se.0.weight"],
                        sd[f"{pfx}.bottleneck.0.se.0.bias"]))
    z = torch.sigmoid(F.linear(z, sd[f"{pfx}.bottleneck.0.se.2.weight"],
                               sd[f"{pfx}.bottleneck.0.se.2.bias"]))
    cat = cat * z[:, :, None, None]
    y = F.conv2d(cat, sd[f"{pfx}.bottleneck.1.weight"], sd[f"{pfx}.bottleneck.1.bias"], padding=1)
    y = F.group_norm(y, g, sd[f"{pfx}.bottleneck.2.weight"], sd[f"{pfx}.bottleneck.2.bias"], 1e-5)
    if keep_mask is not None:
        y = y * (keep_mask.to(y.dtype) / (1.0 - arch.droprate))[:, :, None, None]
    feat = y
    out = F.conv2d(feat, sd[f"{pfx}.head.1.weight"])
    return out, feat


def forward(sd, x, arch=RESNET101, training=False, keep_mask=None, update_stats=False):
    """SegModel.forward: (shallow, deep, logits, feat)."""
    shallow, deep = trunk(sd, x, arch, training, update_stats)
    out, feat = aspp_head(sd, deep, arch, keep_mask)
    return shallow, deep, out, feat


def forward_fixed_masks(sd, x, arch, masks, keep_mask=None):
    """SegModel.forward in train mode (batch-statistics BN) with EVERY switch pinned to a given pattern: the stem / bottleneck /
    ASPP-branch / SE ReLUs become multiplications with 0/1 masks and the 3x3/2 max-pool a gather with given indices
    (`masks`: "layer0", "pool_idx", "layer{l}.{b}.{1,2,3}", "aspp.{b}", "se" -- the patterns observed on the device under
    test).  The network's gradients are discontinuous where a pre-activation crosses zero or two pool candidates tie; with
    the switches pinned the function is smooth around the operating point, so the gradients of ALL layers -- through the stem,
    the max-pool, the strided / dilated stage transitions and the ASPP chain in one backward pass -- can be compared
    elementwise with a tight bound (cf. bottleneck_fixed_masks).  Same reference lines as trunk / aspp_head."""
    y = F.conv2d(x, sd["layer0.0.weight"], stride=2, padding=3)
    y = _bn(sd, "layer0.1", y, True) * masks["layer0"]
    idx = masks["pool_idx"]
    y = y.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    shallow = None
    inplanes = arch.stem
    for li in range(4):
        planes, stride, dil = arch.planes[li], arch.strides[li], arch.dilations[li]
        for bi in range(arch.layers[li]):
            first = bi == 0
            has_down = first and (stride != 1 or inplanes != planes * arch.expansion or dil in (2, 4))
            pfx = f"layer{li + 1}.{bi}"
            y = bottleneck_fixed_masks(sd, pfx, y, stride if first else 1, dil, has_down,
                                       (masks[pfx + ".1"], masks[pfx + ".2"], masks[pfx + ".3"]))
            inplanes = planes * arch.expansion
        if li == 1:
            shallow = y
    deep = y
    g = arch.gn_groups
    branches = []
    for b in range(1 + len(arch.aspp_dilations)):
        w, bias = sd[f"final.conv2d_list.{b}.0.weight"], sd[f"final.conv2d_list.{b}.0.bias"]
        d = 1 if b == 0 else arch.aspp_dilations[b - 1]
        z = F.conv2d(deep, w, bias) if b == 0 else F.conv2d(deep, w, bias, padding=d, dilation=d)
        z = F.group_norm(z, g, sd[f"final.conv2d_list.{b}.1.weight"], sd[f"final.conv2d_list.{b}.1.bias"], 1e-5)
        branches.append(z * masks[f"aspp.{b}"])
    cat = torch.cat(branches, 1)
    pooled = cat.mean(dim=(2, 3))
    z = F.linear(pooled, sd["final.bottleneck.0.se.0.weight"], sd["final.bottleneck.0.se.0.bias"]) * masks["se"]
    z = torch.sigmoid(F.linear(z, sd["final.bottleneck.0.se.2.weight"], sd["final.bottleneck.0.se.2.bias"]))
    cat = cat * z[:, :, None, None]
    y = F.conv2d(cat, sd["final.bottleneck.1.weight"], sd["final.bottleneck.1.bias"], padding=1)
    y = F.group_norm(y, g, sd["final.bottleneck.2.weight"], sd["final.bottleneck.2.bias"], 1e-5)
    if keep_mask is not None:
        y = y * (keep_mask.to(y.dtype) / (1.0 - arch.droprate))[:, :, None, None]
    return shallow, deep, F.conv2d(y, sd["final.head.1.weight"]), y


def state_shapes(arch=RESNET101):
    """Ordered {key: (shape, kind)} in the reference's state_dict order.
    kind in {conv, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, bias, gn_w, gn_b, lin, head}."""
    out = {}

    def bn(pfx, c):
        out[pfx + ".weight"] = ((c,), "bn_w")
        out[pfx + ".bias"] = ((c,), "bn_b")
        out[pfx + ".running_mean"] = ((c,), "bn_rm")
        out[pfx + ".running_var"] = ((c,), "bn_rv")
        out[pfx + ".num_batches_tracked"] = ((), "bn_nbt")

    out["layer0.0.weight"] = ((arch.stem, 3, 7, 7), "conv")
    bn("layer0.1", arch.stem)
    inplanes = arch.stem
    for li in range(4):
        planes, stride, dil = arch.planes[li], arch.strides[li], arch.dilations[li]
        for bi in range(arch.layers[li]):
            p = f"layer{li + 1}.{bi}"
            out[p + ".conv1.weight"] = ((planes, inplanes, 1, 1), "conv")
            bn(p + ".bn1", planes)
            out[p + ".conv2.weight"] = ((planes, planes, 3, 3), "conv")
            bn(p + ".bn2", planes)
            out[p + ".conv3.weight"] = ((planes * arch.expansion, planes, 1, 1), "conv")
            bn(p + ".bn3", planes * arch.expansion)
            if bi == 0 and (stride != 1 or inplanes != planes * arch.expansion or dil in (2, 4)):
                out[p + ".downsample.0.weight"] = ((planes * arch.expansion, inplanes, 1, 1), "conv")
                bn(p + ".downsample.1", planes * arch.expansion)
            inplanes = planes * arch.expansion
    w = arch.aspp_width
    nb = 1 + len(arch.aspp_dilations)
    for b in range(nb):
        k = 1 if b == 0 else 3
        out[f"final.conv2d_list.{b}.0.weight"] = ((w, inplanes, k, k), "conv")
        out[f"final.conv2d_list.{b}.0.bias"] = ((w,), "bias")
        out[f"final.conv2d_list.{b}.1.weight"] = ((w,), "gn_w")
        out[f"final.conv2d_list.{b}.1.bias"] = ((w,), "gn_b")
    cat = w * nb
    out["final.bottleneck.0.se.0.weight"] = ((cat // arch.se_reduction, cat), "lin")
    out["final.bottleneck.0.se.0.bias"] = ((cat // arch.se_reduction,), "bias")
    out["final.bottleneck.0.se.2.weight"] = ((cat, cat // arch.se_reduction), "lin")
    out["final.bottleneck.0.se.2.bias"] = ((cat,), "bias")
    out["final.bottleneck.1.weight"] = ((w, cat, 3, 3), "conv")
    out["final.bottleneck.1.bias"] = ((w,), "bias")
    out["final.bottleneck.2.weight"] = ((w,), "gn_w")
    out["final.bottleneck.2.bias"] = ((w,), "gn_b")
    out["final.head.1.weight"] = ((arch.n_classes, w, 1, 1), "head")
    return out
