"""Deterministic, name-hashed weights for parity runs (test infrastructure only).

The golden G-model capture loads these into the *reference* SegModel in the build
container; the GPU parity tests load the same values into the build's model, so
no 260 MB checkpoint has to be committed (pin G-model).
Values depend only on (key name, shape, kind): a CPU torch.Generator seeded with
crc32(name).  Fully random ReLU/BN stacks are chaotic in train mode (a 1e-6 relative
weight perturbation moved logits by 5e-3 with unit bn3 gains), which would turn every
parity check into a noise measurement; the last BN of each bottleneck therefore gets a
gain of ~0.2, which brings the amplification down to ~100x.
"""
import zlib

import torch

from .deeplab import RESNET101, state_shapes


def _gen(name):
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return g


def fill(name, shape, kind):
    g = _gen(name)
    if kind == "bn_nbt":
        return torch.zeros((), dtype=torch.int64)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if kind == "conv":
        fan_in = shape[1] * shape[2] * shape[3]
        return r * (2.0 / fan_in) ** 0.5
    if kind == "head":
        return r * 0.05
    if kind == "lin":
        return r * (1.0 / shape[1]) ** 0.5
    if kind == "bn_w" and name.endswith(".bn3.weight"):
        # damped residual branch, as in a trained network: keeps train-mode (batch-stat BN)
        # forward passes well conditioned, so fp32 summation-order noise stays ~1e-5 on logits
        return 0.2 * (1.0 + 0.1 * r)
    if kind in ("bn_w", "gn_w"):
        return 1.0 + 0.1 * r
    if kind in ("bn_b", "gn_b", "bias", "bn_rm"):
        return 0.1 * r
    if kind == "bn_rv":
        return 1.0 + 0.2 * torch.rand(shape, generator=g, dtype=torch.float32)
    raise ValueError(kind)


def state_dict(arch=RESNET101):
    return {k: fill(k, shp, kind) for k, (shp, kind) in state_shapes(arch).items()}
