"""Frozen source->target image translator (inference only), state-dict compatible with the reference's
`ImgEncoder` / `ImgDecoder` (G5/model/model_noaux.py:80-117 built from G5/model/model_util.py:21-61,121-174).

It runs every step under no_grad right before the scoped path (warm_up.py:235-237; SURVEY section 8f "next"
row 1).  Everything but the residual adds runs on the HIP kernels: the convolutions on the implicit-GEMM kernels with
the reflection padding folded into their tap addressing, the decoder's nearest-neighbour x2 upsampling folded into the
loader of the conv that follows (the 4x larger tensor is never written), tanh in the last conv's epilogue
(diga_conv_options_t of the `_opts` conv entry points), InstanceNorm (+ReLU) on the GroupNorm kernels with one group per channel.  With autograd
enabled the same modules fall back to explicit pad / upsample / tanh ops around the conv (the translator is frozen on
the DiGA path; its GAN training is out of scope).
Keys: `model.<i>.conv.{weight,bias}` for plain blocks, `model.<i>.model.<j>.model.<k>.conv.{weight,bias}`
for the residual stacks.
"""
import torch
import torch.nn as nn

from diga_amd.model import norm as dn
from diga_amd.model.conv import DigaConv2d


class FoldedUpsample(nn.Upsample):
    """nn.Upsample(scale_factor=2) (nearest) that, under no_grad, only TAGS its input: the ConvBlock that follows reads
    the source tensor through the conv kernel's upsampling input map."""

    def forward(self, x):
        if torch.is_grad_enabled() or self.mode != "nearest" or float(self.scale_factor) != 2.0 or not x.is_cuda:
            return super().forward(x)
        y = x.view_as(x)                 # a fresh tensor object on the same memory: the tag must not outlive this call on `x`
        y._diga_up_shift = 1
        return y


class DigaInstanceNorm2d(nn.InstanceNorm2d):
    """nn.InstanceNorm2d(affine=False, eps=1e-5) (+ fused ReLU) = GroupNorm with one group per channel and unit affine,
    on the HIP GroupNorm kernels (per-(image, channel) statistics over H*W)."""

    def forward(self, x, relu=False):
        if self.affine or self.track_running_stats:
            raise NotImplementedError("DigaInstanceNorm2d: plain instance statistics only (the translator's setting)")
        c = x.shape[1]
        ones = getattr(self, "_ones", None)
        if ones is None or ones.device != x.device or ones.numel() != c:
            self._ones = ones = torch.ones(c, dtype=torch.float32, device=x.device)
            self._zeros = torch.zeros(c, dtype=torch.float32, device=x.device)
        return dn._GnFn.apply(x, ones, self._zeros, c, self.eps, bool(relu), None, None)

_ACT = {"relu": lambda: nn.ReLU(inplace=True), "lrelu": lambda: nn.LeakyReLU(0.2, inplace=True),
        "tanh": nn.Tanh, "none": lambda: None}


class ConvBlock(nn.Module):
    """pad -> conv (no implicit padding) -> [InstanceNorm] -> [activation]."""

    def __init__(self, cin, cout, k, stride=1, padding=0, norm="none", activation="relu", pad_type="zero"):
        super().__init__()
        self.pad = nn.ReflectionPad2d(padding) if pad_type == "reflect" else nn.ZeroPad2d(padding)
        if norm not in ("in", "none"):
            raise NotImplementedError(f"norm '{norm}' is not used by the translator")
        self.norm = DigaInstanceNorm2d(cout) if norm == "in" else None
        self.activation = _ACT[activation]()
        self.act_name = activation
        self.conv = DigaConv2d(cin, cout, k, stride, bias=True)
        self.padding, self.pad_type = padding, pad_type

    def forward(self, x):
        up = int(getattr(x, "_diga_up_shift", 0))
        # inference: padding / upsampling / tanh ride inside the conv kernel (image-like 3-channel inputs take the
        # im2col stem path, which pads with zeros only: they keep the explicit -- tiny -- reflection pad)
        fold = (not torch.is_grad_enabled()) and x.is_cuda and self.conv.in_channels >= 8 and self.pad_type in ("reflect", "zero")
        if fold:
            conv = self.conv
            saved = conv.padding
            conv.padding = (self.padding, self.padding)          # geometry of pad + conv in one kernel
            try:
                tanh = self.act_name == "tanh" and self.norm is None
                x = conv(x, opts=(1 if self.pad_type == "reflect" and self.padding > 0 else 0, up, 1 if tanh else 0))
            finally:
                conv.padding = saved
            return self._tail(x, tanh)
        if up:
            x = nn.functional.interpolate(x, scale_factor=2.0, mode="nearest")
        return self._tail(self.conv(self.pad(x)), False)

    def _tail(self, x, tanh_done):
        if self.norm is not None:
            if self.act_name in ("relu", "none"):
                return self.norm(x, relu=self.act_name == "relu")          # ReLU fused into the norm's apply pass
            x = self.norm(x).clone()                                          # (custom-Function output: no in-place op on it)
        return x if (self.activation is None or tanh_done) else self.activation(x)


class ResBlock(nn.Module):
    def __init__(self, dim, norm, activation, pad_type):
        super().__init__()
        self.model = nn.Sequential(ConvBlock(dim, dim, 3, 1, 1, norm, activation, pad_type),
                                   ConvBlock(dim, dim, 3, 1, 1, norm, "none", pad_type))

    def forward(self, x):
        return x + self.model(x)


class ResBlocks(nn.Module):
    def __init__(self, n, dim, norm="in", activation="relu", pad_type="zero"):
        super().__init__()
        self.model = nn.Sequential(*[ResBlock(dim, norm, activation, pad_type) for _ in range(n)])

    def forward(self, x):
        return self.model(x)


class ImgEncoder(nn.Module):
    def __init__(self, input_dim=3, dim=64, n_downsample=2, n_res=4, activ="relu", norm="in", pad_type="reflect"):
        super().__init__()
        layers = [ConvBlock(input_dim, dim, 7, 1, 3, norm, activ, pad_type)]
        for _ in range(n_downsample):
            layers.append(ConvBlock(dim, 2 * dim, 4, 2, 1, norm, activ, pad_type))
            dim *= 2
        layers.append(ResBlocks(n_res, dim, norm, activ, pad_type))
        self.model = nn.Sequential(*layers)
        self.output_dim = dim

    def forward(self, x):
        return self.model(x)


class ImgDecoder(nn.Module):
    def __init__(self, dim=256, output_dim=3, n_upsample=2, n_res=4, norm="in", activ="relu", pad_type="reflect"):
        super().__init__()
        layers = [ResBlocks(n_res, dim, norm, activ, pad_type)]
        for _ in range(n_upsample):
            layers += [FoldedUpsample(scale_factor=2), ConvBlock(dim, dim // 2, 5, 1, 2, "in", activ, pad_type)]
            dim //= 2
        layers.append(ConvBlock(dim, output_dim, 7, 1, 3, "none", "tanh", pad_type))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)
