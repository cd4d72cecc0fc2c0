"""Oracle (CPU, test infrastructure only): confusion-matrix mIoU.

Restates G5/util/metrics.py:26-68 (runningScore) with numpy.
Pinned by tests/golden (G-miou).
"""
import numpy as np


def confusion(gt, pred, n_classes=19):
    """metrics.py:32-37: rows = ground truth, cols = prediction, gt outside [0,n) dropped."""
    gt = np.asarray(gt).reshape(-1).astype(np.int64)
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    keep = (gt >= 0) & (gt < n_classes)
    idx = gt[keep] * n_classes + pred[keep]
    return np.bincount(idx, minlength=n_classes * n_classes).reshape(n_classes, n_classes)


def scores(hist):
    """metrics.py:46-65."""
    hist = hist.astype(np.float64)
    diag = np.diag(hist)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = diag.sum() / hist.sum()
        acc_cls = np.nanmean(diag / hist.sum(axis=1))
        iu = diag / (hist.sum(axis=1) + hist.sum(axis=0) - diag)
        miou = np.nanmean(iu)
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return {"acc": acc, "acc_cls": acc_cls, "fwavacc": fwavacc, "miou": miou, "iu": iu}
