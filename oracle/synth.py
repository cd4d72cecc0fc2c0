"""Synthetic inputs shared by the golden generator, the parity tests and bench.py's
cpu_baseline leg (test infrastructure only).  SURVEY section 8d: everything comes from
a CPU torch.Generator so the same seed gives the same tensors on every box.
"""
import torch


def gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def block_labels(g, b, h, w, block, n_classes=19, ignore_frac=0.02):
    """Labels made of block x block patches of one class, `ignore_frac` pixels set to 255."""
    hb, wb = (h + block - 1) // block, (w + block - 1) // block
    coarse = torch.randint(0, n_classes, (b, hb, wb), generator=g)
    lab = coarse.repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :h, :w].contiguous()
    if ignore_frac > 0:
        drop = torch.rand((b, h, w), generator=g) < ignore_frac
        lab[drop] = 255
    return lab.to(torch.int64)


def warmup_batch(seed, b, h, w, block=32):
    """(x, x_aug, rec_s2t, labels) of the warm-up step."""
    g = gen(seed)
    x = torch.rand((b, 3, h, w), generator=g) * 2.0 - 1.0
    labels = block_labels(g, b, h, w, block)
    rec = torch.tanh(torch.randn((b, 3, h, w), generator=g))
    x_aug = x + 0.1 * torch.randn((b, 3, h, w), generator=g)
    return x, x_aug, rec, labels


def selftrain_batch(seed, b, h, w, block=32, redraw=0.3):
    """Adds target images, their augmented view and offline pseudo-labels (C4)."""
    x, x_aug, rec, labels = warmup_batch(seed, b, h, w, block)
    g = gen(seed + 7919)
    t = torch.rand((b, 3, h, w), generator=g) * 2.0 - 1.0
    t_aug = t + 0.1 * torch.randn((b, 3, h, w), generator=g)
    t_lab = block_labels(g, b, h, w, block)
    noisy = block_labels(g, b, h, w, block, ignore_frac=0.0)
    hb, wb = (h + block - 1) // block, (w + block - 1) // block
    flip = (torch.rand((b, hb, wb), generator=g) < redraw)
    flip = flip.repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :h, :w]
    pseudo = torch.where(flip, noisy, t_lab)
    return x, x_aug, rec, labels, t, t_aug, pseudo


def checksum(t):
    """Order-sensitive float64 checksum of a tensor."""
    v = t.detach().to(torch.float64).reshape(-1)
    k = torch.arange(1, v.numel() + 1, dtype=torch.float64)
    return float((v * torch.cos(k * 0.37)).sum())
