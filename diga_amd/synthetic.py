"""Synthetic batches of the benchmark configurations: everything is drawn from a
CPU torch.Generator seeded with 1234 + rank, then copied to the device, so a configuration is the
same tensor on every box."""
import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def block_labels(g, b, h, w, block, n_classes=19, ignore_frac=0.02):
    hb, wb = (h + block - 1) // block, (w + block - 1) // block
    coarse = torch.randint(0, n_classes, (b, hb, wb), generator=g)
    lab = coarse.repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :h, :w].contiguous()
    if ignore_frac > 0:
        lab[torch.rand((b, h, w), generator=g) < ignore_frac] = 255
    return lab.to(torch.int64)


def warmup_batch(seed, b, h, w, block=32, device="cpu"):
    """(x, x_aug, rec_s2t, labels): source crops, their colour-augmented view, the translated image
    (stand-ins for the kornia / translator outputs that feed the path) and block-structured labels."""
    g = _gen(seed)
    x = torch.rand((b, 3, h, w), generator=g) * 2.0 - 1.0
    labels = block_labels(g, b, h, w, block)
    rec = torch.tanh(torch.randn((b, 3, h, w), generator=g))
    x_aug = x + 0.1 * torch.randn((b, 3, h, w), generator=g)
    return tuple(t.to(device) for t in (x, x_aug, rec, labels))


def selftrain_batch(seed, b, h, w, block=32, redraw=0.3, device="cpu"):
    """warmup_batch + (target images, augmented target view, offline pseudo-labels)."""
    x, x_aug, rec, labels = warmup_batch(seed, b, h, w, block)
    g = _gen(seed + 7919)
    t = torch.rand((b, 3, h, w), generator=g) * 2.0 - 1.0
    t_aug = t + 0.1 * torch.randn((b, 3, h, w), generator=g)
    t_lab = block_labels(g, b, h, w, block)
    noisy = block_labels(g, b, h, w, block, ignore_frac=0.0)
    hb, wb = (h + block - 1) // block, (w + block - 1) // block
    flip = (torch.rand((b, hb, wb), generator=g) < redraw)
    flip = flip.repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :h, :w]
    pseudo = torch.where(flip, noisy, t_lab)
    return tuple(v.to(device) for v in (x, x_aug, rec, labels, t, t_aug, pseudo))
