"""Oracle (CPU, test infrastructure only): the SegFormer all-MLP decode head, restated functionally on a state dict.

Follows /root/reference/domain_adaptation/GTA5/model/networks/segformer_head.py:
  MLP.forward :19-22 (flatten -> Linear), SegFormerHead.__init__ :58-70 (four MLPs to 768 channels, `linear_fuse` =
  ConvModule(4*768 -> 768, k=1, norm BN), `linear_pred` = Conv2d(768 -> classes, k=1)), SegFormerHead.forward :137-165
  (embed, resize to c1's grid with bilinear / align_corners=False, cat [c4, c3, c2, c1], fuse, Dropout2d, predict),
  resize :472-488 (F.interpolate).
Evaluated in the REFERENCE's order (the 4*768-channel concatenation is formed), plain torch ops in the dtype of the state dict.

Pin: tests/golden/segformer_head.npz is a capture of the reference's `SegFormerHead` class itself (tools/gen_golden.py::
gen_segformer_head imports the reference file).  mmcv is absent from this image; what the file takes from it is `ConvModule`
(used once, for `linear_fuse`) and `normal_init` (initialisation, unused by the capture).  The capture stands `ConvModule` in as
Conv2d(bias=False) -> BatchNorm2d -> ReLU -- mmcv 1.x's documented behaviour for `norm_cfg=dict(type='BN')` with the default
`bias='auto'`, `order=('conv', 'norm', 'act')` and `act_cfg=dict(type='ReLU')` -- so THAT block is pinned to mmcv's documentation,
not to mmcv's code ("parity unpinned" for the ConvModule block; everything else in the capture is the reference's own code).
Dropout2d is the identity in the capture (p = 0) and here (`keep_mask` carries a drawn mask when a test wants one).
"""
import zlib

import torch
import torch.nn.functional as F

EMBED = 768          # segformer_head.py:50 hard-codes decoder_params = dict(embed_dim=768, ...)


def state_shapes(in_channels=(64, 128, 320, 512), num_classes=19, embed=EMBED):
    """Key -> (shape, kind) in the reference's state-dict order (segformer_head.py:58-70; ConvModule's children are `conv`, `bn`)."""
    out = {}
    for lvl in (4, 3, 2, 1):
        out[f"linear_c{lvl}.proj.weight"] = ((embed, in_channels[lvl - 1]), "lin")
        out[f"linear_c{lvl}.proj.bias"] = ((embed,), "bias")
    out["linear_fuse.conv.weight"] = ((embed, 4 * embed, 1, 1), "conv")
    out["linear_fuse.bn.weight"] = ((embed,), "bn_w")
    out["linear_fuse.bn.bias"] = ((embed,), "bn_b")
    out["linear_fuse.bn.running_mean"] = ((embed,), "bn_rm")
    out["linear_fuse.bn.running_var"] = ((embed,), "bn_rv")
    out["linear_fuse.bn.num_batches_tracked"] = ((), "bn_nbt")
    out["linear_pred.weight"] = ((num_classes, embed, 1, 1), "head")
    out["linear_pred.bias"] = ((num_classes,), "bias")
    return out


def state_dict(in_channels=(64, 128, 320, 512), num_classes=19, embed=EMBED, dtype=torch.float32):
    """Deterministic name-hashed values (the scheme of oracle/detweights.py)."""
    sd = {}
    for k, (shp, kind) in state_shapes(in_channels, num_classes, embed).items():
        g = torch.Generator(device="cpu")
        g.manual_seed(zlib.crc32(("segformer_head." + k).encode()) & 0x7FFFFFFF)
        if kind == "bn_nbt":
            sd[k] = torch.zeros((), dtype=torch.int64)
            continue
        r = torch.randn(shp, generator=g, dtype=torch.float32)
        if kind == "lin":
            v = r * (1.0 / shp[1]) ** 0.5
        elif kind == "conv":
            v = r * (2.0 / shp[1]) ** 0.5
        elif kind == "head":
            v = r * 0.05
        elif kind == "bn_w":
            v = 1.0 + 0.1 * r
        elif kind == "bn_rv":
            v = 1.0 + 0.2 * torch.rand(shp, generator=g, dtype=torch.float32)
        else:
            v = 0.1 * r
        sd[k] = v.to(dtype)
    return sd


def forward(sd, feats, training=True, eps=1e-5, keep_mask=None):
    """feats = [c1, c2, c3, c4] (NCHW) -> (logits, _c_raw, _c).  Train mode uses batch statistics (the running buffers in `sd` are
    left alone; `batch_stats` below gives what they would be updated with)."""
    c1 = feats[0]
    n = c1.shape[0]
    embedded = []
    for lvl in (4, 3, 2, 1):                                    # :140-147
        c = feats[lvl - 1]
        t = F.linear(c.flatten(2).transpose(1, 2), sd[f"linear_c{lvl}.proj.weight"], sd[f"linear_c{lvl}.proj.bias"])
        t = t.permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])
        if lvl != 1:
            t = F.interpolate(t, size=c1.shape[2:], mode="bilinear", align_corners=False)
        embedded.append(t)
    c_raw = torch.cat(embedded, dim=1)                          # :149
    y = F.conv2d(c_raw, sd["linear_fuse.conv.weight"])          # ConvModule: conv (no bias) -> BN -> ReLU
    if training:
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
    else:
        mean, var = sd["linear_fuse.bn.running_mean"], sd["linear_fuse.bn.running_var"]
    xhat = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    fused = torch.relu(xhat * sd["linear_fuse.bn.weight"][None, :, None, None] + sd["linear_fuse.bn.bias"][None, :, None, None])
    x = fused if keep_mask is None else fused * keep_mask[:, :, None, None]      # Dropout2d (:160): per (image, channel) keep / (1 - p)
    logits = F.conv2d(x, sd["linear_pred.weight"], sd["linear_pred.bias"])       # :161
    return logits, c_raw, fused


def batch_stats(sd, feats):
    """(mean, unbiased variance) of the fuse conv's output: what one train-mode forward blends into the running buffers."""
    with torch.no_grad():
        c1 = feats[0]
        n = c1.shape[0]
        embedded = []
        for lvl in (4, 3, 2, 1):
            c = feats[lvl - 1]
            t = F.linear(c.flatten(2).transpose(1, 2), sd[f"linear_c{lvl}.proj.weight"], sd[f"linear_c{lvl}.proj.bias"])
            t = t.permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])
            if lvl != 1:
                t = F.interpolate(t, size=c1.shape[2:], mode="bilinear", align_corners=False)
            embedded.append(t)
        y = F.conv2d(torch.cat(embedded, dim=1), sd["linear_fuse.conv.weight"])
        return y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=True)
