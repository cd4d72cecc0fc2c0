"""Colour-augmentation view of the DiGA scripts on MI355X:

    sdatav_aug = beta * norm(extra_aug(sdatav)) + (1 - beta) * sdatav     G5/train_DiGA_gta2city_warm_up.py:105-111,233

`extra_aug` there is a kornia 0.5.8 pipeline (ColorJitter(0.4, 0.4, 0.2, 0.1, p=0.5) -> RandomGrayscale(p=0.3) ->
RandomGaussianBlur((3,3),(2,2), p=0.8) -> RandomSharpness(0.5, p=0.3)); kornia is not installable on the target boxes.
`ExtraAug` is a drop-in for that nn.Sequential (same call: image batch in, image batch out) and `color_aug_view` is the
whole line in one HBM pass (`diga_color_aug_view`).  kornia's algorithms are restated from its published source
(oracle/coloraug.py cites the files; PARITY UNPINNED against kornia itself); the per-sample random decisions come from a
counter-based generator -- a pure function of (seed, call counter, sample index, draw index): reproducible, no state on
the device, no host sync (kornia's own torch-distribution sampling is not reproduced).
"""
import os
import sys

import numpy as np
import torch

_pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib  # noqa: E402

__all__ = ["ExtraAug", "color_aug_view", "draw_params"]

_U32 = np.uint32


def _mix32(v):
    v = _U32(v)
    with np.errstate(over="ignore"):
        v ^= v >> _U32(16)
        v = _U32(v * _U32(0x85EBCA6B))
        v ^= v >> _U32(13)
        v = _U32(v * _U32(0xC2B2AE35))
        v ^= v >> _U32(16)
    return v


def _uniform01(seed, sample, draw):
    """float32 in [0, 1) from 24 bits of a murmur3-finaliser hash of (seed, sample, draw)."""
    with np.errstate(over="ignore"):
        a = _mix32(_U32(seed) ^ _U32(_U32(0x9E3779B9) * _U32(sample + 1)))
        b = _mix32(_U32(a + _U32(draw)))
    return np.float32(b >> _U32(8)) * np.float32(1.0 / 16777216.0)


def draw_params(seed, batch, brightness=0.4, contrast=0.4, saturation=0.2, hue=0.1, p_jitter=0.5, p_gray=0.3,
                p_blur=0.8, p_sharp=0.3, sharp_max=0.5):
    """([batch, 12] float32 table of diga_color_aug_view, [4] int32 jitter order).  Ranges as kornia 0.5.8 derives them
    from the constructor arguments: factors uniform in [1-b, 1+b] (brightness, contrast, saturation), [-h, h] (hue),
    sharpness uniform in [0, sharp_max]; one order per batch."""
    f32 = np.float32
    lo = np.array([1 - brightness, 1 - contrast, 1 - saturation, -hue], dtype=f32)
    hi = np.array([1 + brightness, 1 + contrast, 1 + saturation, hue], dtype=f32)
    tab = np.zeros((batch, 12), dtype=f32)
    for b in range(batch):
        tab[b, 0] = 1.0 if _uniform01(seed, b, 0) < f32(p_jitter) else 0.0
        for k in range(4):
            tab[b, 4 + k] = f32(lo[k] + (hi[k] - lo[k]) * _uniform01(seed, b, 1 + k))
        tab[b, 1] = 1.0 if _uniform01(seed, b, 5) < f32(p_gray) else 0.0
        tab[b, 2] = 1.0 if _uniform01(seed, b, 6) < f32(p_blur) else 0.0
        tab[b, 3] = 1.0 if _uniform01(seed, b, 7) < f32(p_sharp) else 0.0
        tab[b, 8] = f32(f32(sharp_max) * _uniform01(seed, b, 8))
    order = [0, 1, 2, 3]
    for i in range(3, 0, -1):
        j = min(int(_uniform01(seed, 0xFFFFFF, 3 - i) * f32(i + 1)), i)
        order[i], order[j] = order[j], order[i]
    return tab, np.array(order, dtype=np.int32)


def color_aug_view(x, beta, mean, std, params, order):
    """beta * Normalize(mean, std)(extra_aug(x)) + (1 - beta) * x for x [B,3,H,W] on the GPU; `params`, `order` from
    draw_params (or any table of the documented layout)."""
    _lib.require_gpu(x)
    xc = _lib.contiguous(x.detach(), torch.float32)
    b, c, h, w = xc.shape
    if c != 3:
        raise ValueError("color_aug_view: images must have 3 channels")
    tab = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float32)).to(xc.device, non_blocking=True)
    if tuple(tab.shape) != (b, 12):
        raise ValueError(f"color_aug_view: params must be [{b}, 12]")
    out = torch.empty_like(xc)
    ordr = np.ascontiguousarray(order, dtype=np.int32)
    m3 = np.ascontiguousarray(mean, dtype=np.float32)
    s3 = np.ascontiguousarray(std, dtype=np.float32)
    if ordr.shape != (4,) or m3.shape != (3,) or s3.shape != (3,):
        raise ValueError("color_aug_view: order must have 4 entries, mean / std 3")
    _lib.call("diga_color_aug_view", _lib.ptr(xc), _lib.ptr(out), _lib.ptr(tab), ordr.ctypes.data, b, h, w, float(beta),
              m3.ctypes.data, s3.ctypes.data, _lib.stream())
    return out


class ExtraAug(torch.nn.Module):
    """Drop-in for the reference's `extra_aug` nn.Sequential (warm_up.py:105-111): y = extra_aug(x).  Every call draws
    fresh per-sample parameters from (seed, call counter)."""

    def __init__(self, brightness=0.4, contrast=0.4, saturation=0.2, hue=0.1, p_jitter=0.5, p_gray=0.3, p_blur=0.8,
                 p_sharp=0.3, sharpness=0.5, seed=0):
        super().__init__()
        self.kw = dict(brightness=brightness, contrast=contrast, saturation=saturation, hue=hue, p_jitter=p_jitter,
                       p_gray=p_gray, p_blur=p_blur, p_sharp=p_sharp, sharp_max=sharpness)
        self.seed = int(seed)
        self.calls = 0

    def next_params(self, batch):
        tab, order = draw_params((self.seed * 1000003 + self.calls) & 0xFFFFFFFF, batch, **self.kw)
        self.calls += 1
        return tab, order

    def forward(self, x):
        tab, order = self.next_params(x.shape[0])
        return color_aug_view(x, 1.0, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), tab, order)

    def view(self, x, beta, mean, std):
        """beta * Normalize(mean, std)(self(x)) + (1 - beta) * x in one kernel (the whole of warm_up.py:233)."""
        tab, order = self.next_params(x.shape[0])
        return color_aug_view(x, beta, mean, std, tab, order)
