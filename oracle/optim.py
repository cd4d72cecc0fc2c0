"""Oracle (CPU, test infrastructure only): EMA teacher, poly LR, duplicate-aware SGD.

Restates
  * G5/util/utils.py:93-116  create_teacher_params / update_teacher_params
  * G5/util/utils.py:32-41   poly_lr_scheduler / adjust_learning_rate
  * torch.optim.SGD(momentum, weight_decay), single-tensor path, driven with the
    duplicate-laden 1x group that G5/model/model_noaux.py:48-63 yields
    (SURVEY App. A-9).
Pinned by tests/golden (G-ema, G-sgd, KAT-4).
"""
import torch


def ema_alpha(iteration, stage0=True, mean=False, replace=False):
    """utils.py:105-112."""
    if stage0:
        return min(1.0 - 1.0 / (iteration + 1), 0.999)
    if mean:
        return 0.9
    if replace:
        return 0.0
    return 0.999


def ema_update(teacher_params, student_params, iteration, **kw):
    """utils.py:113-115: t <- a*t + (1-a)*s over parameters only (buffers untouched)."""
    a = ema_alpha(iteration, **kw)
    with torch.no_grad():
        for t, s in zip(teacher_params, student_params):
            t.copy_(a * t + (1.0 - a) * s)
    return a


def poly_lr(base_lr, it, max_iter, power=0.9):
    """utils.py:32-33."""
    return base_lr * ((1.0 - float(it) / max_iter) ** power)


def sgd_step_dup(params, grads, bufs, mults, lrs, momentum=0.9, weight_decay=5e-4,
                 first_step=False):
    """One optimizer.step() of torch.optim.SGD (foreach=False) when tensor i occurs
    mults[i] times in its group.  Each occurrence is a sequential micro-step:
        d = g + wd*p ; buf = d (fresh) or momentum*buf + d ; p -= lr*buf
    torch 2.10 allocates a fresh buffer for EVERY occurrence on the first step
    (state is only written back after the loop) and shares one buffer afterwards.
    In-place on params/bufs (lists of fp32 tensors)."""
    with torch.no_grad():
        for p, g, b, k, lr in zip(params, grads, bufs, mults, lrs):
            for _ in range(int(k)):
                d = g + weight_decay * p
                if first_step:
                    b.copy_(d)
                else:
                    b.mul_(momentum).add_(d)
                p.add_(b, alpha=-lr)
