"""Oracle (CPU, test infrastructure only): two-scale validation pass.

Restates G5/train_DiGA_gta2city_warm_up.py:346-359 / G5/evaluate_val.py:73-93: logits of the image and of its
half-size bilinear (align_corners) downscale, both upsampled to label size, element-wise max, argmax,
confusion matrix.  Built from oracle.losses.upsample_bilinear_ac (pinned by tests/golden/upsample.npz) and
oracle.metrics.confusion (pinned by tests/golden/miou.npz).
"""
import torch

from .losses import upsample_bilinear_ac
from .metrics import confusion


def two_scale_prediction(pred, pred_ds, size):
    up = torch.max(upsample_bilinear_ac(pred_ds, size), upsample_bilinear_ac(pred, size))
    return up.argmax(dim=1), up


def evaluate_two_scale(forward, images, labels, ds_size=None, n_classes=19):
    """forward(x) -> logits.  Returns (prediction [N,H,W], confusion [K,K], fused logits)."""
    H, W = labels.shape[-2:]
    ds_size = ds_size or (images.shape[-2] // 2, images.shape[-1] // 2)
    image_ds = upsample_bilinear_ac(images, ds_size)
    pred, fused = two_scale_prediction(forward(images), forward(image_ds), (H, W))
    return pred, confusion(labels.numpy(), pred.numpy(), n_classes), fused
