"""runningScore with the reference's interface (G5/util/metrics.py:26-68); the confusion matrix is
accumulated on the GPU by diga_confusion_matrix (integer atomics), the IoU arithmetic stays numpy
float64 exactly as in the reference."""
import os
import sys

import numpy as np
import torch

_pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib  # noqa: E402

# Cityscapes train-id names, printed with the per-class IoU exactly as the reference does
label = ("road sidewalk building wall fence pole light sign vegetation terrain sky person rider car truck bus "
         "train motorcycle bycycle").split()


def _summary(conf):
    """Accuracy, class accuracy, IoU, frequency-weighted IoU of a confusion matrix (rows = ground truth)."""
    tp = np.diag(conf)
    per_gt, per_pred = conf.sum(axis=1), conf.sum(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = tp / (per_gt + per_pred - tp)
        freq = per_gt / conf.sum()
        return (tp.sum() / conf.sum(), np.nanmean(tp / per_gt), (freq[freq > 0] * iou[freq > 0]).sum(),
                np.nanmean(iou), iou)


class runningScore(object):
    def __init__(self, n_classes, device=None, verbose=True):
        self.n_classes = n_classes
        self.verbose = verbose
        self._device = torch.device(device if device is not None else "cuda")
        self._hist = None

    def _dev_hist(self):
        if self._hist is None:
            self._hist = torch.zeros(self.n_classes * self.n_classes, dtype=torch.int64, device=self._device)
        return self._hist

    def update(self, label_trues, label_preds):
        """gt / prediction batches: numpy arrays or tensors (any device); values outside [0, n) in gt are ignored."""
        gt = torch.as_tensor(np.asarray(label_trues) if not torch.is_tensor(label_trues) else label_trues)
        pr = torch.as_tensor(np.asarray(label_preds) if not torch.is_tensor(label_preds) else label_preds)
        gt = gt.to(self._device, torch.int64).contiguous()
        pr = pr.to(self._device, torch.int64).contiguous()
        if gt.numel() != pr.numel():
            raise ValueError("runningScore.update: gt and prediction differ in size")
        hist = self._dev_hist()
        _lib.call("diga_confusion_matrix", _lib.ptr(gt), _lib.ptr(pr), _lib.ptr(hist), gt.numel(), self.n_classes,
                  _lib.stream())

    @property
    def confusion_matrix(self):
        if self._hist is None:
            return np.zeros((self.n_classes, self.n_classes))
        return self._hist.cpu().numpy().reshape(self.n_classes, self.n_classes).astype(np.float64)

    def get_scores(self):
        acc, acc_cls, fwavacc, mean_iu, iu = _summary(self.confusion_matrix)
        if self.verbose:
            for i in range(min(self.n_classes, len(label))):
                print('===>' + label[i] + ':' + str(iu[i]))
        cls_iu = dict(zip(range(self.n_classes), iu))
        return {'Overall Acc: \t': acc, 'Mean Acc : \t': acc_cls, 'FreqW Acc : \t': fwavacc,
                'Mean IoU : \t': mean_iu}, cls_iu

    def reset(self):
        self._hist = None
