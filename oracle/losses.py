"""Oracle (CPU, test infrastructure only): per-pixel losses of the DiGA hot path.

Restates, in vectorised PyTorch-CPU, the arithmetic of
  * G5/util/loss.py:48-62   cross_entropy2d
  * G5/util/loss.py:125-143 distillation_loss
  * nn.Upsample(bilinear, align_corners=True) as used at
    G5/train_DiGA_gta2city_warm_up.py:173-176,267-268,274
Pinned against reference captures in tests/golden (G-ce, G-distil, KAT-1/2).
"""
import torch
import torch.nn.functional as F

IGNORE = 255


def bilinear_taps(n_in, n_out, dtype=torch.float32):
    """Source taps of an align_corners=True bilinear resize along one axis.

    dst index d reads src = d*(n_in-1)/(n_out-1); returns (i0, i1, w1) with the
    value (1-w1)*x[i0] + w1*x[i1].  n_out == 1 maps to src 0 (torch semantics).
    """
    d = torch.arange(n_out, dtype=dtype)
    scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    src = d * torch.tensor(scale, dtype=dtype)
    i0 = src.floor().to(torch.int64).clamp_(0, n_in - 1)
    i1 = (i0 + 1).clamp_(max=n_in - 1)
    w1 = src - i0.to(dtype)
    return i0, i1, w1


def upsample_bilinear_ac(x, size):
    """[N,C,h,w] -> [N,C,H,W], bilinear, align_corners=True (explicit gather form)."""
    H, W = size
    h, w = x.shape[-2:]
    y0, y1, wy = bilinear_taps(h, H, x.dtype)
    x0, x1, wx = bilinear_taps(w, W, x.dtype)
    top = x[..., y0, :]
    bot = x[..., y1, :]
    rows = top + (bot - top) * wy[:, None]
    left = rows[..., x0]
    right = rows[..., x1]
    return left + (right - left) * wx


def cross_entropy2d(logits, target, size_average=True):
    """G5/util/loss.py:48-62.  Sum over non-255 pixels of -log p[target], divided
    by ALL pixels N*H*W (the reference's `mask = target >= 0` is always true,
    SURVEY App. A-1)."""
    n, c, h, w = logits.shape
    logp = F.log_softmax(logits, dim=1)
    valid = target != IGNORE
    idx = torch.where(valid, target, torch.zeros_like(target)).unsqueeze(1)
    picked = logp.gather(1, idx).squeeze(1)
    loss = -(picked * valid.to(logp.dtype)).sum()
    if size_average:
        loss = loss / float(n * h * w)
    return loss


def ohem_cross_entropy(score, target, thresh=0.7, min_kept=100000, ignore_label=IGNORE):
    """OhemCrossEntropy._ohem_forward, G5/util/loss.py:91-109, restated without the sort: over valid pixels
    p_t = softmax(score)[target]; kth = the min(min_kept, n_valid-1)-th smallest p_t; threshold = max(kth, thresh);
    the loss is the mean CE of the valid pixels with p_t < threshold.  Scores at another resolution are first
    upsampled (bilinear, align_corners), :92-96.  Returns (loss, kept mask [N,H,W], threshold)."""
    if tuple(score.shape[-2:]) != tuple(target.shape[-2:]):
        score = upsample_bilinear_ac(score, tuple(target.shape[-2:]))
    min_kept = max(1, int(min_kept))
    logp = F.log_softmax(score, dim=1)
    valid = target != ignore_label
    idx = torch.where(valid, target, torch.zeros_like(target)).unsqueeze(1)
    ce = -logp.gather(1, idx).squeeze(1)
    pt = F.softmax(score, dim=1).gather(1, idx).squeeze(1)
    pv = pt[valid]
    if pv.numel() == 0:
        return score.sum() * float("nan"), torch.zeros_like(valid), float("nan")
    k = min(min_kept, pv.numel() - 1)
    kth = torch.kthvalue(pv.detach().reshape(-1), k + 1).values       # 0-based rank k
    threshold = max(float(kth), float(thresh))
    kept = valid & (pt < threshold)
    return ce[kept].mean(), kept, threshold


def distillation_loss(teacher_out, student_out, scale=0.5):
    """G5/util/loss.py:125-143.  Views are the two halves of the batch; teacher
    of view 0 supervises student of view 1 (weight 1) and teacher of view 1
    supervises student of view 0 (weight `scale`); each term is a mean over
    B*H*W of sum_c -q*log_softmax(s).  Teacher is detached."""
    q = F.softmax(teacher_out, dim=1).detach()
    q0, q1 = q.chunk(2)
    s0, s1 = student_out.chunk(2)
    t01 = (-(q0 * F.log_softmax(s1, dim=1)).sum(1)).mean()
    t10 = (-(q1 * F.log_softmax(s0, dim=1)).sum(1)).mean()
    return t01 + scale * t10


def ce_grad(logits, target):
    """Closed-form d cross_entropy2d / d logits (used to cross-check autograd)."""
    n, c, h, w = logits.shape
    p = F.softmax(logits, dim=1)
    valid = (target != IGNORE)
    onehot = F.one_hot(torch.where(valid, target, torch.zeros_like(target)), c)
    onehot = onehot.permute(0, 3, 1, 2).to(p.dtype)
    return (p - onehot) * valid.unsqueeze(1).to(p.dtype) / float(n * h * w)


def distill_grad(teacher_out, student_out, scale=0.5):
    """Closed-form d distillation_loss / d student."""
    q = F.softmax(teacher_out, dim=1)
    q0, q1 = q.chunk(2)
    s0, s1 = student_out.chunk(2)
    b, c, h, w = s0.shape
    denom = float(b * h * w)
    g0 = scale * (F.softmax(s0, dim=1) - q1) / denom
    g1 = (F.softmax(s1, dim=1) - q0) / denom
    return torch.cat([g0, g1], 0)


def warmup_losses_lowres(stu_lr, tea_lr, labels, lambda_seg=1.0, lambda_distil=0.5,
                         scale=0.5):
    """The loss block of the warm-up step at the LOW-RES boundary
    (warm_up.py:267-282,299): upsample both logit stacks to label resolution,
    CE on the first half of the student stack, distillation on both stacks.
    Returns (total, ce, distil); differentiable wrt stu_lr."""
    size = labels.shape[-2:]
    b = labels.shape[0]
    stu = upsample_bilinear_ac(stu_lr, size)
    tea = upsample_bilinear_ac(tea_lr, size)
    ce = cross_entropy2d(stu[:b], labels)
    di = distillation_loss(tea, stu, scale)
    return lambda_seg * ce + lambda_distil * di, ce, di
