"""Losses of the DiGA hot path on MI355X -- same names and call signatures as the reference's
`util/loss.py` (G5/util/loss.py:48-62 cross_entropy2d, :125-143 distillation_loss), computed by
the HIP kernels of libdiga_hip.so.  No CPU fallback.

Extra (not in the reference): `upsample_ce_distill` / `upsample_ce`, the same losses taken at
the low-res logit boundary with the bilinear(align_corners) upsampling fused in, which the
build's own step driver uses instead of materialising [2B,19,H,W] tensors.
"""
import os
import sys

import torch

_pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg) not in sys.path:          # drop-in use: only .../diga_amd is on sys.path
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib  # noqa: E402

__all__ = ["cross_entropy2d", "distillation_loss", "upsample_ce_distill", "upsample_ce", "upsample_ce_distill_aux",
           "OhemCrossEntropy"]


def _f32c(t):
    return _lib.contiguous(t, torch.float32)


def _apply_upstream(grad, grad_out):
    """grad *= grad_out on device, without a host sync (no traffic when grad_out == 1)."""
    go = _f32c(grad_out.reshape(1))
    _lib.call("diga_scale_inplace", _lib.ptr(grad), _lib.ptr(go), grad.numel(), _lib.stream())
    return grad


class _CrossEntropy2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        _lib.require_gpu(logits, target)
        x = _f32c(logits.detach())
        t = _lib.contiguous(target, torch.int64)
        n, c, h, w = x.shape
        if tuple(t.shape) != (n, h, w):
            raise ValueError(f"cross_entropy2d: target shape {tuple(t.shape)} does not match logits {tuple(x.shape)}")
        grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        nbytes = _lib.lib.diga_loss_workspace_bytes(n * h * w)
        ws = _lib.workspace(nbytes, x.device, "loss")
        _lib.call("diga_ce2d_fwd_bwd", _lib.ptr(x), _lib.ptr(t), _lib.ptr(grad), _lib.ptr(loss), _lib.ptr(ws),
                  ws.numel(), n, c, h, w, 1.0, _lib.stream())
        ctx.grad = grad
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        grad, ctx.grad = ctx.grad, None
        if grad is None:
            raise RuntimeError("cross_entropy2d: backward called twice (the fused gradient is consumed once)")
        return _apply_upstream(grad, grad_out), None


def cross_entropy2d(input, target, weight=None, size_average=True):
    """Per-pixel C-way cross entropy, ignore label 255, normalised by ALL N*H*W pixels
    (reference semantics, G5/util/loss.py:56-61)."""
    if weight is not None:
        raise NotImplementedError("cross_entropy2d: class weights are not used on the DiGA path")
    loss = _CrossEntropy2d.apply(input, target)
    if not size_average:
        n, _, h, w = input.shape
        loss = loss * float(n * h * w)
    return loss


class _OhemCrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, target, ignore_label, thresh, min_kept):
        _lib.require_gpu(score, target)
        x = _f32c(score.detach())
        t = _lib.contiguous(target, torch.int64)
        n, c, h, w = x.shape
        if tuple(t.shape) != (n, h, w):
            raise ValueError(f"OhemCrossEntropy: target shape {tuple(t.shape)} does not match score {tuple(x.shape)}")
        grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        ws = _lib.workspace(_lib.lib.diga_ohem_ce_workspace_bytes(n, h, w), x.device, "ohem")
        _lib.call("diga_ohem_ce_fwd_bwd", _lib.ptr(x), _lib.ptr(t), _lib.ptr(grad), _lib.ptr(loss), _lib.ptr(ws), ws.numel(),
                  n, c, h, w, int(ignore_label), float(thresh), int(min_kept), 1.0, _lib.stream())
        ctx.grad = grad
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        grad, ctx.grad = ctx.grad, None
        if grad is None:
            raise RuntimeError("OhemCrossEntropy: backward called twice (the fused gradient is consumed once)")
        return _apply_upstream(grad, grad_out), None, None, None, None


class OhemCrossEntropy(torch.nn.Module):
    """Hard-pixel cross entropy with the reference's constructor and call signature (G5/util/loss.py:65-122; the
    Synthia / semi-supervised scripts' `seg_loss`).  Scores at another resolution than the labels are upsampled
    (bilinear, align_corners) first, as the reference does; the selection threshold comes from an exact on-device
    radix select instead of a sort (`diga_ohem_ce_fwd_bwd`)."""

    def __init__(self, ignore_label=255, thres=0.7, min_kept=100000, weight=None):
        super().__init__()
        if weight is not None:
            raise NotImplementedError("OhemCrossEntropy: class weights are not used on the DiGA path")
        self.thresh = thres
        self.min_kept = max(1, min_kept)
        self.ignore_label = ignore_label

    def forward(self, score, target):
        if tuple(score.shape[-2:]) != tuple(target.shape[-2:]):
            score = torch.nn.functional.interpolate(score, size=tuple(target.shape[-2:]), mode="bilinear",
                                                    align_corners=True)
        return _OhemCrossEntropy.apply(score, target, self.ignore_label, self.thresh, self.min_kept)


class _Distillation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, teacher, student, scale):
        _lib.require_gpu(teacher, student)
        t = _f32c(teacher.detach())
        s = _f32c(student.detach())
        if t.shape != s.shape or s.dim() != 4 or s.shape[0] % 2:
            raise ValueError(f"distillation_loss: need two equal [2B,C,H,W] stacks, got {tuple(t.shape)} / {tuple(s.shape)}")
        b2, c, h, w = s.shape
        grad = torch.empty_like(s) if ctx.needs_input_grad[1] else None
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        nbytes = _lib.lib.diga_loss_workspace_bytes(b2 * h * w)
        ws = _lib.workspace(nbytes, s.device, "loss")
        _lib.call("diga_distill_fwd_bwd", _lib.ptr(t), _lib.ptr(s), _lib.ptr(grad), _lib.ptr(loss), _lib.ptr(ws),
                  ws.numel(), b2, c, h, w, float(scale), 1.0, _lib.stream())
        ctx.grad = grad
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        grad, ctx.grad = ctx.grad, None
        if grad is None:
            raise RuntimeError("distillation_loss: backward called twice (the fused gradient is consumed once)")
        return None, _apply_upstream(grad, grad_out), None


def distillation_loss(teacher_out, student_out, scale=0.5):
    """Symmetric cross-view soft-target CE (G5/util/loss.py:125-143); differentiable wrt student_out only."""
    return _Distillation.apply(teacher_out, student_out, scale)


class _UpsampleCeDistill(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stu_lr, tea_lr, labels, lambda_seg, lambda_distil, scale):
        _lib.require_gpu(stu_lr, tea_lr, labels)
        s = _f32c(stu_lr.detach())
        t = _f32c(tea_lr.detach())
        lab = _lib.contiguous(labels, torch.int64)
        b2, c, h, w = s.shape
        b, H, W = lab.shape
        if b2 != 2 * b or t.shape != s.shape:
            raise ValueError(f"upsample_ce_distill: logits {tuple(s.shape)}/{tuple(t.shape)} vs labels {tuple(lab.shape)}")
        grad = torch.empty_like(s)
        losses = torch.empty(2, dtype=torch.float32, device=s.device)
        nbytes = _lib.lib.diga_upsample_loss_workspace_bytes(b2, c, h, w)
        ws = _lib.workspace(nbytes, s.device, "upsample_loss")
        _lib.call("diga_upsample_ce_distill_fwd_bwd", _lib.ptr(s), _lib.ptr(t), _lib.ptr(lab), _lib.ptr(grad),
                  _lib.ptr(losses), _lib.ptr(ws), ws.numel(), b, c, h, w, H, W, float(lambda_seg),
                  float(lambda_distil), float(scale), _lib.stream())
        ctx.grad = grad
        total = float(lambda_seg) * losses[0] + float(lambda_distil) * losses[1]
        ctx.mark_non_differentiable(losses)
        return total, losses

    @staticmethod
    def backward(ctx, grad_total, _grad_losses):
        grad, ctx.grad = ctx.grad, None
        if grad is None:
            raise RuntimeError("upsample_ce_distill: backward called twice")
        return _apply_upstream(grad, grad_total), None, None, None, None, None


def upsample_ce_distill(stu_lr, tea_lr, labels, lambda_seg=1.0, lambda_distil=0.5, scale=0.5):
    """total = lambda_seg*cross_entropy2d(up(stu)[:B], labels) + lambda_distil*distillation_loss(up(tea), up(stu), scale)
    with up = bilinear align_corners upsampling to labels' size (warm_up.py:267-282,299), fused.
    Returns (total, ce, distil); total is differentiable wrt stu_lr."""
    total, losses = _UpsampleCeDistill.apply(stu_lr, tea_lr, labels, lambda_seg, lambda_distil, scale)
    return total, losses[0], losses[1]


class _UpsampleCe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_lr, labels, lambda_seg):
        _lib.require_gpu(logits_lr, labels)
        s = _f32c(logits_lr.detach())
        lab = _lib.contiguous(labels, torch.int64)
        n, c, h, w = s.shape
        nb, H, W = lab.shape
        if nb != n:
            raise ValueError("upsample_ce: batch mismatch")
        grad = torch.empty_like(s)
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        nbytes = _lib.lib.diga_upsample_loss_workspace_bytes(n, c, h, w)
        ws = _lib.workspace(nbytes, s.device, "upsample_loss")
        _lib.call("diga_upsample_ce_fwd_bwd", _lib.ptr(s), _lib.ptr(lab), _lib.ptr(grad), _lib.ptr(loss), _lib.ptr(ws),
                  ws.numel(), n, c, h, w, H, W, float(lambda_seg), _lib.stream())
        ctx.grad = grad
        return loss.reshape(()) * float(lambda_seg)

    @staticmethod
    def backward(ctx, grad_out):
        grad, ctx.grad = ctx.grad, None
        if grad is None:
            raise RuntimeError("upsample_ce: backward called twice")
        return _apply_upstream(grad, grad_out), None, None


def upsample_ce(logits_lr, labels, lambda_seg=1.0):
    """lambda_seg * cross_entropy2d(up(logits_lr), labels), upsampling fused (self_training.py:343-351)."""
    return _UpsampleCe.apply(logits_lr, labels, lambda_seg)


def upsample_ce_distill_aux(stu_main, stu_aux, tea_main, tea_aux, labels, lambda_seg=1.0, lambda_distil=0.5,
                            lambda_aux=0.1, scale=0.5):
    """Loss block of the two-headed (HRNet-OCR) student of the semi-supervised tree
    (semi-supervised_segmentation/train_DiGA_semiseg_warm_up.py:239-263,282):

        loss_semseg = CE(up(main)[:B], y) + lambda_aux * CE(up(aux)[:B], y)
        loss_distil = distill(up(tea_main), up(stu_main)) + lambda_aux * distill(up(tea_aux), up(stu_aux))
        total       = lambda_seg * loss_semseg + lambda_distil * loss_distil

    taken at the low-res logit boundary with the upsampling fused: two launches of the fused loss block, one per head
    (the network itself is outside the scoped path; its four logit maps are the inputs here).  Returns
    (total, loss_semseg, loss_distil); differentiable wrt both student maps."""
    t_m, ce_m, di_m = upsample_ce_distill(stu_main, tea_main, labels, lambda_seg, lambda_distil, scale)
    t_a, ce_a, di_a = upsample_ce_distill(stu_aux, tea_aux, labels, lambda_seg, lambda_distil, scale)
    la = float(lambda_aux)
    return t_m + la * t_a, ce_m + la * ce_a, di_m + la * di_a
