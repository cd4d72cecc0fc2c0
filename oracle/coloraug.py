"""Oracle (CPU, test infrastructure only): the colour-augmentation view of the DiGA scripts,

    sdatav_aug = beta * norm(extra_aug(sdatav)) + (1 - beta) * sdatav        (G5/train_DiGA_gta2city_warm_up.py:105-111,233)

with extra_aug = kornia 0.5.8's ColorJitter(0.4, 0.4, 0.2, 0.1, p=0.5) -> RandomGrayscale(p=0.3) ->
RandomGaussianBlur((3,3),(2,2), p=0.8) -> RandomSharpness(0.5, p=0.3) and norm = util/utils.py:141-156 Normalize.

PARITY UNPINNED: kornia (requirements.txt:15, kornia==0.5.8) is a third-party dependency that is absent from
/root/reference and not installable here (no network), and the reference holds no test or golden vector for this view.
This file restates kornia 0.5.8's published algorithms (kornia/enhance/adjust.py, kornia/color/hsv.py, kornia/color/gray.py,
kornia/filters/gaussian.py, kornia/enhance/adjust.py::sharpness, kornia/augmentation/augmentation.py) as deterministic
functions of EXPLICIT per-sample parameters; kornia's own parameter sampling (torch distributions on its generator) is
not reproduced -- the build draws parameters from a counter-based generator of its own (params_from_seed below, restated
bit for bit by the device code).  What is pinned: Normalize and the beta blend (reference code, RNG-free).

Quirk kept from the reference: extra_aug is fed the mean/std-NORMALISED image, and kornia's brightness / contrast clamp to
[0, 1] -- negative values are cut.  The restatement follows the code, not the intent.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi
GRAY_W = (0.299, 0.587, 0.114)


def rgb_to_hsv(img, eps=1e-6):
    """kornia.color.rgb_to_hsv (0.5.8): h in [0, 2 pi), s, v.  img [3,H,W]."""
    maxc, _ = img.max(0)
    maxc_mask = img == maxc.unsqueeze(0)
    _, max_idx = ((maxc_mask.cumsum(0) == 1) & maxc_mask).max(0)
    minc = img.min(0)[0]
    v = maxc
    deltac = maxc - minc
    s = deltac / (v + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = (maxc.unsqueeze(0) - img).unbind(0)
    h = torch.stack([bc - gc, 2.0 * deltac + rc - bc, 4.0 * deltac + gc - rc], 0)
    h = torch.gather(h, 0, max_idx.unsqueeze(0)).squeeze(0)
    h = h / deltac
    h = (h / 6.0) % 1.0
    return torch.stack([TWO_PI * h, s, v], 0)


def hsv_to_rgb(hsv):
    """kornia.color.hsv_to_rgb (0.5.8)."""
    h = hsv[0] / TWO_PI
    s, v = hsv[1], hsv[2]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    p = v * (1.0 - s)
    q = v * (1.0 - f * s)
    t = v * (1.0 - (1.0 - f) * s)
    hi = hi.long()
    table = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), 0)
    idx = torch.stack([hi, hi + 6, hi + 12], 0)
    return torch.gather(table, 0, idx)


def jitter_op(img, op, value):
    """One ColorJitter step on [3,H,W]: op 0 brightness (additive value-1, clamp), 1 contrast (multiplicative, clamp),
    2 saturation (HSV, s*value clamped), 3 hue (HSV, h + value*2 pi, fmod)."""
    if op == 0:
        return torch.clamp(img + (value - 1.0), 0.0, 1.0)
    if op == 1:
        return torch.clamp(img * value, 0.0, 1.0)
    hsv = rgb_to_hsv(img)
    if op == 2:
        hsv = torch.stack([hsv[0], torch.clamp(hsv[1] * value, 0.0, 1.0), hsv[2]], 0)
    else:
        hsv = torch.stack([torch.fmod(hsv[0] + value * TWO_PI, TWO_PI), hsv[1], hsv[2]], 0)
    return hsv_to_rgb(hsv)


def gaussian_kernel3(sigma=2.0):
    k = torch.tensor([math.exp(-1.0 / (2.0 * sigma * sigma)), 1.0, math.exp(-1.0 / (2.0 * sigma * sigma))], dtype=torch.float32)
    return k / k.sum()


def gaussian_blur3(img, sigma=2.0):
    """kornia.filters.gaussian_blur2d(kernel (3,3), sigma (2,2), border_type='reflect') on [3,H,W]."""
    k1 = gaussian_kernel3(sigma)
    k2 = torch.outer(k1, k1)
    x = F.pad(img.unsqueeze(0), (1, 1, 1, 1), mode="reflect")
    return F.conv2d(x, k2.reshape(1, 1, 3, 3).repeat(3, 1, 1, 1), groups=3)[0]


def sharpness(img, factor):
    """kornia.enhance.sharpness (0.5.8): 3x3 smoothing kernel [[1,1,1],[1,5,1],[1,1,1]]/13 on the interior (the one-pixel
    border keeps the input), then blend smooth + (input - smooth) * factor (clamped only outside (0, 1))."""
    k = torch.tensor([[1.0, 1.0, 1.0], [1.0, 5.0, 1.0], [1.0, 1.0, 1.0]]) / 13.0
    deg = F.conv2d(img.unsqueeze(0), k.reshape(1, 1, 3, 3).repeat(3, 1, 1, 1), groups=3)[0]
    res = img.clone()
    res[:, 1:-1, 1:-1] = deg
    if factor == 0.0:
        return res
    if factor == 1.0:
        return img
    out = res + (img - res) * factor
    return out if 0.0 < factor < 1.0 else torch.clamp(out, 0.0, 1.0)


def extra_aug(x, params):
    """x [B,3,H,W] fp32.  params: dict of per-sample arrays
       jitter [B] bool, factors [B,4] (brightness, contrast, saturation, hue), order [4] (one permutation per batch, as
       kornia draws it), gray [B] bool, blur [B] bool, sharp [B] bool, sharp_factor [B]."""
    out = []
    for b in range(x.shape[0]):
        img = x[b]
        if params["jitter"][b]:
            for op in params["order"]:
                img = jitter_op(img, int(op), float(params["factors"][b][int(op)]))
        if params["gray"][b]:
            g = GRAY_W[0] * img[0] + GRAY_W[1] * img[1] + GRAY_W[2] * img[2]
            img = torch.stack([g, g, g], 0)
        if params["blur"][b]:
            img = gaussian_blur3(img)
        if params["sharp"][b]:
            img = sharpness(img, float(params["sharp_factor"][b]))
        out.append(img)
    return torch.stack(out)


def normalize(x, mean, std):
    """util/utils.py:141-156 Normalize: per channel (x - mean) / std."""
    m = torch.as_tensor(mean, dtype=x.dtype).view(1, -1, 1, 1)
    s = torch.as_tensor(std, dtype=x.dtype).view(1, -1, 1, 1)
    return (x - m) / s


def color_aug_view(x, params, beta, mean, std):
    """warm_up.py:233."""
    return beta * normalize(extra_aug(x, params), mean, std) + (1.0 - beta) * x


# ---- the build's own parameter generator (counter-based; restated bit for bit by csrc/coloraug.hip) --------------
def _mix32(v):
    """32-bit finaliser of splitmix / murmur3 (uint32 numpy arithmetic)."""
    v = np.uint32(v)
    with np.errstate(over="ignore"):
        v ^= v >> np.uint32(16)
        v = np.uint32(v * np.uint32(0x85EBCA6B))
        v ^= v >> np.uint32(13)
        v = np.uint32(v * np.uint32(0xC2B2AE35))
        v ^= v >> np.uint32(16)
    return v


def uniform01(seed, sample, draw):
    """float32 in [0, 1): 24 high bits of mix32(mix32(seed ^ 0x9E3779B9 * (sample + 1)) + draw)."""
    with np.errstate(over="ignore"):
        a = _mix32(np.uint32(seed) ^ np.uint32(np.uint32(0x9E3779B9) * np.uint32(sample + 1)))
        b = _mix32(np.uint32(a + np.uint32(draw)))
    return np.float32(b >> np.uint32(8)) * np.float32(1.0 / 16777216.0)


def params_from_seed(seed, batch, brightness=0.4, contrast=0.4, saturation=0.2, hue=0.1, p_jitter=0.5, p_gray=0.3,
                     p_blur=0.8, p_sharp=0.3, sharp_max=0.5):
    """Per-sample parameters.  Draw indices: 0 jitter?, 1..4 factors, 5 gray?, 6 blur?, 7 sharp?, 8 sharpness factor;
    the jitter order is drawn once per batch from sample index 0xFFFFFF (draws 0..2, Fisher-Yates)."""
    f32 = np.float32
    P = {"jitter": [], "factors": [], "gray": [], "blur": [], "sharp": [], "sharp_factor": []}
    lo = np.array([1 - brightness, 1 - contrast, 1 - saturation, -hue], dtype=f32)
    hi = np.array([1 + brightness, 1 + contrast, 1 + saturation, hue], dtype=f32)
    for b in range(batch):
        P["jitter"].append(bool(uniform01(seed, b, 0) < f32(p_jitter)))
        P["factors"].append([f32(lo[k] + (hi[k] - lo[k]) * uniform01(seed, b, 1 + k)) for k in range(4)])
        P["gray"].append(bool(uniform01(seed, b, 5) < f32(p_gray)))
        P["blur"].append(bool(uniform01(seed, b, 6) < f32(p_blur)))
        P["sharp"].append(bool(uniform01(seed, b, 7) < f32(p_sharp)))
        P["sharp_factor"].append(f32(f32(sharp_max) * uniform01(seed, b, 8)))
    order = [0, 1, 2, 3]
    for i in range(3, 0, -1):
        j = int(uniform01(seed, 0xFFFFFF, 3 - i) * f32(i + 1))
        j = min(j, i)
        order[i], order[j] = order[j], order[i]
    P["order"] = order
    P["factors"] = np.array(P["factors"], dtype=f32)
    P["sharp_factor"] = np.array(P["sharp_factor"], dtype=f32)
    return P
