"""SegFormer all-MLP decode head on the HIP kernels -- host mirror of G5/model/networks/segformer_head.py:12-165 (`MLP`,
`SegFormerHead`; the reference copies it from NVlabs/SegFormer, needs mmcv, and wires it into nothing: SURVEY section 2.1).

Same constructor arguments, same state-dict keys (`linear_c{1..4}.proj.{weight,bias}`, `linear_fuse.conv.weight`,
`linear_fuse.bn.*` -- mmcv's ConvModule names --, `linear_pred.{weight,bias}`), same result: per stage a Linear embedding to 768
channels, bilinear resize (align_corners=False) to the 1/4-scale grid, concatenation [c4, c3, c2, c1], 1x1 fuse conv without bias
-> BatchNorm (trainable affine) -> ReLU, Dropout2d, 1x1 prediction conv (:137-165).

MI355X-first evaluation order.  The reference materialises the 3072-channel concatenation at 1/4 scale (16 crops of 768x768:
7.2 GB in fp32, 2.8 TFLOP for the fuse conv).  Resize acts per channel, the fuse conv per pixel: they commute, and fuse o Linear
is ONE matrix per stage.  So per stage i the weights are folded first (W'_i = Wf[:, slice_i] W_i, 768 x C_i; the biases folded into
one 768-vector), the folded 1x1 conv runs on the stage's OWN grid (58 + 29 + 18 + 7 GFLOP instead of 2.8 TFLOP), and one bandwidth
kernel adds the three coarse maps into the fine one (diga_pyramid_sum_fwd / _bwd, csrc/pyramid.hip).  Gradients reach W_i and Wf
through the fold (two tiny GEMMs per stage, autograd), exact in real arithmetic; in fp32 the result differs from the reference's
order by rounding only (tests/test_gpu_segformer_head.py holds it to the oracle, oracle/segformer_head.py).

Second output.  The reference returns `(logits, _c_raw)` with `_c_raw` the 3072-channel concatenation (:149,164).  Nothing in the
repository reads it; forming it is the 7.2 GB this design avoids.  `return_raw=False` (default) returns the fused 768-channel
feature `_c` (post BatchNorm + ReLU, what a centroid / feature consumer of the DiGA steps would take) in its place;
`return_raw=True` also builds `_c_raw` the reference's way (stock resize + cat) and returns it.
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

_pkg = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib  # noqa: E402
from diga_amd.model.conv import INLINE_WGRAD, DigaConv2d, _Conv2dFn  # noqa: E402
from diga_amd.model.norm import DigaTrainableBatchNorm2d, _SmallLinearFn, nhwc  # noqa: E402

_ONE, _ZERO = (1, 1), (0, 0)


def _pointwise(x, w2d, bias=None):
    """1x1 convolution of an NCHW-shaped tensor with a [Cout, Cin] matrix (any autograd tensor) on the implicit-GEMM kernels."""
    # (INLINE_WGRAD: w2d may be a non-leaf -- its gradient is read by the next backward node, not by the optimizer after the join)
    return _Conv2dFn.apply(x, w2d[:, :, None, None], bias, _ONE, _ZERO, _ONE, None, INLINE_WGRAD)


def _fold(wf, w):
    """wf [E, E'] @ w [E', Cin] -> [E, Cin] on the same kernels (exact-fp32 MFMA or the split-bf16 arithmetic, whichever the
    process selected): wf enters as the 'image' (one row per output channel), w^T as the 1x1 weight.  Differentiable in both."""
    y = _pointwise(wf.t()[None, :, :, None], w.t())           # NCHW-shaped [1, Cin, E, 1]
    return y[0, :, :, 0].t()


class _PyramidSumFn(torch.autograd.Function):
    """fine + bias + sum_k resize(coarse_k), written over `fine` (the caller passes a fresh conv output nobody else reads and must
    not use it afterwards)."""
    last_stats = None

    @staticmethod
    def forward(ctx, fine, bias, *coarse):
        _lib.require_gpu(fine)
        f = fine.detach().permute(0, 2, 3, 1)
        if not (f.is_contiguous() and f.dtype == torch.float32):
            raise RuntimeError("SegFormerHead: the finest map must be a dense NHWC fp32 buffer")
        n, h, w, c = f.shape
        cs = [nhwc(t.detach()) for t in coarse]
        if len(cs) > 3 or any(t.shape[0] != n or t.shape[3] != c for t in cs):
            raise RuntimeError("SegFormerHead: at most three coarse maps of the fine map's batch and channel count")
        b = None if bias is None else bias.detach().float().contiguous()
        by_ratio = {(h // t.shape[1], w // t.shape[2]): t for t in cs if h % t.shape[1] == 0 and w % t.shape[2] == 0}
        _PyramidSumFn.last_stats = None
        if (len(cs) == 3 and h % 16 == 0 and w % 16 == 0 and c % 32 == 0 and n * (c // 32) < 65536
                and sorted(by_ratio) == [(2, 2), (4, 4), (8, 8)]):
            # the SegFormer geometry: tiled kernel, and the BatchNorm behind it gets its statistics from this pass
            stats = torch.empty((n * h * w // 64, 3, c), dtype=torch.float32, device=f.device)
            _lib.call("diga_pyramid_sum_fwd3", _lib.ptr(f), h, w, _lib.ptr(b), _lib.ptr(by_ratio[(2, 2)]), _lib.ptr(by_ratio[(4, 4)]),
                      _lib.ptr(by_ratio[(8, 8)]), _lib.ptr(stats), n, c, _lib.stream())
            _PyramidSumFn.last_stats = (stats, 64)              # (read by the caller right after apply: forward's ctx is gone under no_grad)
        else:
            args = []
            for k in range(3):
                args += [_lib.ptr(cs[k]), cs[k].shape[1], cs[k].shape[2]] if k < len(cs) else [None, 0, 0]
            _lib.call("diga_pyramid_sum_fwd", _lib.ptr(f), h, w, _lib.ptr(b), *args, n, c, _lib.stream())
        ctx.shapes = [tuple(t.shape) for t in cs]
        ctx.has_bias = bias is not None
        # the result lives in `fine`'s buffer (a conv output that no backward reads: convs save their input and weight); it is handed
        # on as a new tensor over that memory -- autograd's in-place protocol (mark_dirty) refuses outputs of custom Functions
        return f.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        gn = g.permute(0, 2, 3, 1)
        if not (gn.is_contiguous() and gn.dtype == torch.float32):
            gn = gn.float().contiguous()
        n, h, w, c = gn.shape
        db = None
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = torch.empty(c, dtype=torch.float32, device=gn.device)
            ws = _lib.workspace(_lib.lib.diga_norm_workspace_bytes(n * h * w, 1, c), gn.device, "norm")
            _lib.call("diga_colsum_nhwc", _lib.ptr(gn), c, _lib.ptr(db), n * h * w, c, _lib.ptr(ws), ws.numel(), _lib.stream())
        outs = []
        want = [ctx.needs_input_grad[2 + k] for k in range(len(ctx.shapes))]
        by_ratio = {(h // s[1], w // s[2]): k for k, s in enumerate(ctx.shapes) if h % s[1] == 0 and w % s[2] == 0}
        if (len(ctx.shapes) == 3 and all(want) and h % 16 == 0 and w % 16 == 0 and c % 16 == 0 and n * (c // 16) < 65536
                and sorted(by_ratio) == [(2, 2), (4, 4), (8, 8)]):
            # the SegFormer geometry (1/8, 1/16, 1/32 maps under a 1/4 map): one pass over the fine gradient for all three
            ds = [torch.empty(s, dtype=torch.float32, device=gn.device) for s in ctx.shapes]
            _lib.call("diga_pyramid_sum_bwd3", _lib.ptr(gn), h, w, _lib.ptr(ds[by_ratio[(2, 2)]]), _lib.ptr(ds[by_ratio[(4, 4)]]),
                      _lib.ptr(ds[by_ratio[(8, 8)]]), n, c, _lib.stream())
            return (g, db, *[d.permute(0, 3, 1, 2) for d in ds])
        for k, shp in enumerate(ctx.shapes):
            if not ctx.needs_input_grad[2 + k]:
                outs.append(None)
                continue
            ds = torch.empty(shp, dtype=torch.float32, device=gn.device)
            _lib.call("diga_pyramid_sum_bwd", _lib.ptr(gn), h, w, _lib.ptr(ds), shp[1], shp[2], n, c, _lib.stream())
            outs.append(ds.permute(0, 3, 1, 2))
        return (g, db, *outs)


class MLP(nn.Module):
    """Linear Embedding (segformer_head.py:12-22): [N, C, H, W] -> [N, H*W, embed_dim]."""

    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)

    def forward(self, x):
        y = _pointwise(x, self.proj.weight, self.proj.bias)    # NCHW-shaped over an NHWC buffer: the token matrix is a view of it
        return y.flatten(2).transpose(1, 2)


class ConvModule(nn.Module):
    """The mmcv block as `linear_fuse` uses it (segformer_head.py:63-68): conv without bias (mmcv: bias='auto' is off when a norm
    follows) -> BatchNorm2d with trainable affine -> ReLU.  Attribute names are mmcv's, so checkpoints load."""

    def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None):
        super().__init__()
        if kernel_size != 1 or (norm_cfg or {}).get("type", "BN") not in ("BN", "SyncBN"):
            raise NotImplementedError("ConvModule: the 1x1 conv + BN + ReLU form of SegFormerHead.linear_fuse only")
        self.conv = DigaConv2d(in_channels, out_channels, kernel_size, bias=False)
        self.bn = DigaTrainableBatchNorm2d(out_channels)
        self.activate = nn.ReLU(inplace=True)
        self.init_weights()

    def init_weights(self):
        """mmcv's ConvModule initialises itself in its constructor (mmcv 1.x `ConvModule.init_weights`: `kaiming_init(self.conv,
        a=0, nonlinearity='relu')` = kaiming NORMAL, mode fan_out; `constant_init(self.norm, 1, bias=0)`), which is what a head built
        from scratch by segformer_head.py:63-68 starts from -- std = sqrt(2 / out_channels) = 0.051 for the 3072 -> 768 fuse conv,
        not nn.Conv2d's default kaiming-uniform (std 0.010): behind the BatchNorm that is a 25x different effective step size."""
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(self.bn.weight, 1.0)
        nn.init.constant_(self.bn.bias, 0.0)

    def forward(self, x):
        return self.bn(self.conv(x), relu=True)


class SegFormerHead(nn.Module):
    """SegFormer: Simple and Efficient Design for Semantic Segmentation with Transformers (segformer_head.py:25-165)."""

    def __init__(self, in_channels=None, channels=None, feature_strides=None, num_classes=None, in_index=None, act_cfg=dict(type='ReLU'),
                 dropout_ratio=0.1, conv_cfg=None, norm_cfg=None, input_transform='multiple_select', align_corners=False,
                 decoder_params=None, return_raw=False, **kwargs):
        super().__init__()
        assert len(feature_strides) == len(in_channels)
        assert min(feature_strides) == feature_strides[0]
        self._init_inputs(in_channels, in_index, input_transform)
        self.channels, self.num_classes, self.dropout_ratio = channels, num_classes, dropout_ratio
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.align_corners, self.feature_strides = align_corners, feature_strides
        self.return_raw = bool(return_raw)
        if input_transform != 'multiple_select' or len(self.in_channels) != 4:
            raise NotImplementedError("SegFormerHead: four selected stage outputs (input_transform='multiple_select')")
        if align_corners:
            raise NotImplementedError("SegFormerHead: align_corners=False (the value every SegFormer config uses)")
        c1, c2, c3, c4 = self.in_channels
        e = (decoder_params or {}).get("embed_dim", 768)        # (the reference hard-codes 768, :50)
        self.embedding_dim = e
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.linear_c4 = MLP(input_dim=c4, embed_dim=e)
        self.linear_c3 = MLP(input_dim=c3, embed_dim=e)
        self.linear_c2 = MLP(input_dim=c2, embed_dim=e)
        self.linear_c1 = MLP(input_dim=c1, embed_dim=e)
        self.linear_fuse = ConvModule(in_channels=e * 4, out_channels=e, kernel_size=1, norm_cfg=dict(type='BN', requires_grad=True))
        self.linear_pred = DigaConv2d(e, num_classes, kernel_size=1)

    def _init_inputs(self, in_channels, in_index, input_transform):
        if input_transform is not None:
            assert input_transform in ['resize_concat', 'multiple_select']
            assert isinstance(in_channels, (list, tuple)) and isinstance(in_index, (list, tuple))
            assert len(in_channels) == len(in_index)
            self.in_channels = sum(in_channels) if input_transform == 'resize_concat' else list(in_channels)
        else:
            assert isinstance(in_channels, int) and isinstance(in_index, int)
            self.in_channels = in_channels
        self.input_transform, self.in_index = input_transform, in_index

    def init_weights(self):
        nn.init.normal_(self.linear_pred.weight, mean=0, std=0.01)
        if self.linear_pred.bias is not None:
            nn.init.constant_(self.linear_pred.bias, 0)

    def _transform_inputs(self, inputs):
        return [inputs[i] for i in self.in_index]

    def raw_features(self, feats):
        """`_c_raw` the reference's way (:140-149): embed, resize every map to c1's grid, concatenate [c4, c3, c2, c1]."""
        c1 = feats[0]
        n = c1.shape[0]
        outs = []
        for lin, c in ((self.linear_c4, feats[3]), (self.linear_c3, feats[2]), (self.linear_c2, feats[1]), (self.linear_c1, c1)):
            t = lin(c).permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])
            if t.shape[2:] != c1.shape[2:]:
                t = F.interpolate(t, size=c1.shape[2:], mode='bilinear', align_corners=False)
            outs.append(t)
        return torch.cat(outs, dim=1)

    def forward(self, inputs):
        feats = self._transform_inputs(inputs)                  # 1/4, 1/8, 1/16, 1/32
        _lib.require_gpu(feats[0])
        e = self.embedding_dim
        wf = self.linear_fuse.conv.weight                        # [E, 4E, 1, 1], input channels ordered [c4, c3, c2, c1]
        order = ((self.linear_c4, feats[3]), (self.linear_c3, feats[2]), (self.linear_c2, feats[1]), (self.linear_c1, feats[0]))
        maps = []
        for k, (lin, c) in enumerate(order):
            folded = _fold(wf[:, k * e:(k + 1) * e, 0, 0], lin.proj.weight)
            maps.append(_pointwise(c, folded))
        bcat = torch.cat([lin.proj.bias for lin, _ in order])[None]
        bias = _SmallLinearFn.apply(bcat, wf[:, :, 0, 0], None, 0)[0]
        fused = _PyramidSumFn.apply(maps[3], bias, maps[2], maps[1], maps[0])
        stats, _PyramidSumFn.last_stats = _PyramidSumFn.last_stats, None
        if stats is not None:
            fused._diga_bn_partials = stats                    # (what a DigaConv2d with emit_bn_stats hands its BatchNorm)
        _c = self.linear_fuse.bn(fused, relu=True)
        x = self.dropout(_c) if self.dropout is not None else _c
        x = self.linear_pred(x)
        return x, (self.raw_features(feats) if self.return_raw else _c)
