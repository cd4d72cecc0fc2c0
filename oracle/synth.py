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


# ---- a LEARNABLE task (round 6: training-level mIoU evidence, tools/gen_golden.py::gen_trainmiou) ---------------------------
def class_palette(n_classes=19):
    """One BGR colour per class, fixed for all seeds: well-separated points of the [-0.8, 0.8] cube (a 3 x 3 x 3 grid walked in
    an order that puts consecutive class ids far apart)."""
    pts = torch.tensor([[a, b, c] for a in (-0.8, 0.0, 0.8) for b in (-0.8, 0.0, 0.8) for c in (-0.8, 0.0, 0.8)], dtype=torch.float32)
    order = [(7 * i + 3) % 27 for i in range(27)]
    return pts[order][:n_classes].contiguous()


def learnable_batch(seed, b, h, w, block=16, noise=0.15, ignore_frac=0.02):
    """(x, x_aug, rec_s2t, labels) whose labels are a deterministic function of the image content: every block x block patch
    is painted in its class's palette colour + white noise; `rec_s2t` (the 'translated' view) is the same scene under a global
    contrast / brightness change, `x_aug` the colour-augmented view (extra noise).  A network can reach a high mIoU on held-out
    seeds, so a trained model's score says whether training WORKED -- which random labels cannot."""
    g = gen(seed)
    hb, wb = (h + block - 1) // block, (w + block - 1) // block
    coarse = torch.randint(0, 19, (b, hb, wb), generator=g)
    lab = coarse.repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :h, :w].contiguous()
    pal = class_palette()
    clean = pal[lab].permute(0, 3, 1, 2).contiguous()                      # [b, 3, h, w]
    x = clean + noise * torch.randn((b, 3, h, w), generator=g)
    gain = 0.7 + 0.2 * torch.rand((b, 1, 1, 1), generator=g)
    shift = 0.1 * torch.randn((b, 3, 1, 1), generator=g)
    rec = torch.tanh(gain * x + shift)
    x_aug = x + 0.1 * torch.randn((b, 3, h, w), generator=g)
    labels = lab.to(torch.int64)
    if ignore_frac > 0:
        labels[torch.rand((b, h, w), generator=g) < ignore_frac] = 255
    return x, x_aug, rec, labels
