"""Frozen source->target image translator (inference only), state-dict compatible with the reference's
`ImgEncoder` / `ImgDecoder` (G5/model/model_noaux.py:80-117 built from G5/model/model_util.py:21-61,121-174).

It runs every step under no_grad right before the scoped path (warm_up.py:235-237; SURVEY section 8f "next"
row 1).  The convolutions go through the HIP implicit-GEMM kernel (`DigaConv2d`, padding applied by a
reflection pad in front, as the reference does); InstanceNorm / nearest upsampling / tanh are stock ops.
Keys: `model.<i>.conv.{weight,bias}` for plain blocks, `model.<i>.model.<j>.model.<k>.conv.{weight,bias}`
for the residual stacks.
"""
import torch.nn as nn

from diga_amd.model.conv import DigaConv2d

_ACT = {"relu": lambda: nn.ReLU(inplace=True), "lrelu": lambda: nn.LeakyReLU(0.2, inplace=True),
        "tanh": nn.Tanh, "none": lambda: None}


class ConvBlock(nn.Module):
    """pad -> conv (no implicit padding) -> [InstanceNorm] -> [activation]."""

    def __init__(self, cin, cout, k, stride=1, padding=0, norm="none", activation="relu", pad_type="zero"):
        super().__init__()
        self.pad = nn.ReflectionPad2d(padding) if pad_type == "reflect" else nn.ZeroPad2d(padding)
        if norm not in ("in", "none"):
            raise NotImplementedError(f"norm '{norm}' is not used by the translator")
        self.norm = nn.InstanceNorm2d(cout) if norm == "in" else None
        self.activation = _ACT[activation]()
        self.conv = DigaConv2d(cin, cout, k, stride, bias=True)

    def forward(self, x):
        x = self.conv(self.pad(x))
        if self.norm is not None:
            x = self.norm(x)
        return x if self.activation is None else self.activation(x)


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
            layers += [nn.Upsample(scale_factor=2), ConvBlock(dim, dim // 2, 5, 1, 2, "in", activ, pad_type)]
            dim //= 2
        layers.append(ConvBlock(dim, output_dim, 7, 1, 3, "none", "tanh", pad_type))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)
