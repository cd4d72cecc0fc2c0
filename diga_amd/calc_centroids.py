"""Centroid bank and dynamic pseudo-label selection on MI355X -- `Class_Features` with the
reference's attributes and methods (G5/calc_centroids.py:84-180), computed by libdiga_hip.so.

Reference-compatible (list-returning) methods keep the reference's host round trips where its API
forces them; the build's own step driver uses the device-resident fast paths
`update_from_batch` and `consensus_pseudo_labels`, which never synchronise.
"""
import os
import sys

import numpy as np
import torch

_pkg = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_pkg) not in sys.path:
    sys.path.append(os.path.dirname(_pkg))
from diga_amd import _lib  # noqa: E402


def _f32c(t):
    return _lib.contiguous(t, torch.float32)


class Class_Features:
    def __init__(self, numbers=19, feat_dim=256):
        self.class_numbers = numbers
        self.feat_dim = feat_dim
        self.objective_vectors = torch.zeros([self.class_numbers, feat_dim])
        self.objective_vectors_num = torch.zeros([self.class_numbers])
        self.centroid_momentum = 0.0001
        self.valid_classes = list(range(numbers))
        self.min_pixels = 5

    # ---------------------------------------------------------------- state placement
    def _state_on(self, device):
        if self.objective_vectors.device != device or self.objective_vectors.dtype != torch.float32:
            self.objective_vectors = self.objective_vectors.to(device, torch.float32)
        if not self.objective_vectors.is_contiguous():
            self.objective_vectors = self.objective_vectors.contiguous()
        if self.objective_vectors_num.device != device:
            self.objective_vectors_num = self.objective_vectors_num.to(device, torch.float32)
        return self.objective_vectors, self.objective_vectors_num

    # ---------------------------------------------------------------- distance / weights (calc_centroids.py:166-180)
    def _weights(self, feat, want_dist):
        _lib.require_gpu(feat)
        f = _f32c(feat.detach())
        n, d, h, w = f.shape
        cents, _ = self._state_on(f.device)
        k = self.class_numbers
        if cents.shape != (k, d):
            raise ValueError(f"centroids are {tuple(cents.shape)}, features have {d} channels")
        weights = torch.empty((n, k, h, w), dtype=torch.float32, device=f.device)
        neg = torch.empty_like(weights) if want_dist else None
        _lib.call("diga_centroid_softmax_weights", _lib.ptr(f), _lib.ptr(cents.detach()), _lib.ptr(weights),
                  _lib.ptr(neg), n, d, k, h * w, _lib.stream())
        return weights, neg

    def feat_centroid_distance(self, feat):
        return -self._weights(feat, True)[1]

    def get_centroid_weight(self, feat):
        """softmax over classes of -||centroid_k - feat||_2  -> [N,K,h,w]."""
        return self._weights(feat, False)[0]

    def get_centroid_distance(self, feat):
        return self._weights(feat, True)[1]

    # ---------------------------------------------------------------- bilateral consensus (self_training.py:298-304)
    def consensus_pseudo_labels(self, feat, pseudo_prob, return_feat_pseudo=False):
        """Keep the offline pseudo-label only where it equals argmax_k of the upsampled centroid weights."""
        weights = self.get_centroid_weight(feat)
        lab = _lib.contiguous(pseudo_prob, torch.int64)
        n, k, h, w = weights.shape
        H, W = lab.shape[-2:]
        out = torch.empty_like(lab)
        fp = torch.empty_like(lab) if return_feat_pseudo else None
        _lib.call("diga_upsample_argmax_consensus", _lib.ptr(weights), _lib.ptr(lab), _lib.ptr(out), _lib.ptr(fp),
                  n, k, h, w, H, W, _lib.stream())
        return (out, fp) if return_feat_pseudo else out

    # ---------------------------------------------------------------- class means (calc_centroids.py:120-145)
    def _class_sums(self, feat_cls, outputs, labels_lr=None, labels_full=None):
        _lib.require_gpu(feat_cls, outputs)
        f = _f32c(feat_cls.detach())
        o = _f32c(outputs.detach())
        n, d, h, w = f.shape
        k = self.class_numbers
        if o.shape != (n, k, h, w):
            raise ValueError(f"outputs {tuple(o.shape)} do not match features {tuple(f.shape)} / {k} classes")
        H = W = 0
        ll = lf = None
        if labels_lr is not None:
            ll = _f32c(labels_lr.detach().reshape(n, h, w))
        elif labels_full is not None:
            lf = _lib.contiguous(labels_full, torch.int64)
            H, W = lf.shape[-2:]
        sums = torch.empty((n, k, d), dtype=torch.float32, device=f.device)
        counts = torch.empty((n, k), dtype=torch.int32, device=f.device)
        nbytes = _lib.lib.diga_class_mean_workspace_bytes(n, h * w)
        ws = _lib.workspace(nbytes, f.device, "class_mean")
        _lib.call("diga_class_mean_vectors", _lib.ptr(f), _lib.ptr(o), _lib.ptr(ll), _lib.ptr(lf), _lib.ptr(sums),
                  _lib.ptr(counts), _lib.ptr(ws), ws.numel(), n, d, k, h, w, H, W, _lib.stream())
        return sums, counts, h * w

    def calculate_mean_vector(self, feat_cls, outputs, labels_val=None, model=None):
        """Reference API: (list of [D,1,1] mean vectors, list of class ids), image-major / class-minor,
        classes with < 5 member pixels dropped.  Returning Python lists costs one D->H copy of the
        [N,K] counts (the reference pays two .item() syncs per (image, class))."""
        sums, counts, hw = self._class_sums(feat_cls, outputs, labels_lr=labels_val)
        cnt = counts.cpu().numpy()
        vectors, ids = [], []
        for n in range(cnt.shape[0]):
            for t in range(self.class_numbers):
                c = int(cnt[n, t])
                if c == 0 or c < self.min_pixels:
                    continue
                v = (sums[n, t] / float(hw)) / (float(c) / float(hw))
                vectors.append(v.reshape(-1, 1, 1))
                ids.append(t)
        return vectors, ids

    # ---------------------------------------------------------------- centroid update (calc_centroids.py:147-164)
    def _apply(self, sums, counts, hw, min_pixels, mode):
        cents, nums = self._state_on(sums.device)
        n, k, d = sums.shape
        _lib.call("diga_centroid_ema_apply", _lib.ptr(cents), _lib.ptr(nums), _lib.ptr(sums), _lib.ptr(counts),
                  n, k, d, hw, float(self.centroid_momentum), int(min_pixels), int(mode), _lib.stream())

    def update_objective_SingleVector(self, id, vector, name='moving_average', start_mean=True):
        """Reference API, one (class, vector) at a time."""
        dev = self.objective_vectors.device if self.objective_vectors.is_cuda else torch.device("cuda")
        v = torch.as_tensor(np.asarray(vector) if not torch.is_tensor(vector) else vector)
        v = v.detach().to(dev, torch.float32).reshape(-1)
        self._state_on(dev)
        if start_mean and float(self.objective_vectors_num[id]) < 100:
            name = 'mean'
        if name not in ('moving_average', 'mean'):
            raise NotImplementedError('no such updating way of objective vectors {}'.format(name))
        k, d = self.class_numbers, v.numel()
        sums = torch.zeros((1, k, d), dtype=torch.float32, device=dev)
        sums[0, id] = v
        counts = torch.zeros((1, k), dtype=torch.int32, device=dev)
        counts[0, id] = 1
        self._apply(sums, counts, 1, 1, 0 if name == 'moving_average' else 1)

    def update_from_batch(self, feat_cls, outputs, labels_full=None, labels_lr=None, name='moving_average'):
        """Fast path of `calculate_mean_vector` + the sequential `update_objective_SingleVector(...,
        start_mean=False)` loop (self_training.py:327-341): everything stays on the device, no sync.
        labels_full are [N,H,W] int64 labels, nearest-downsampled in the kernel."""
        sums, counts, hw = self._class_sums(feat_cls, outputs, labels_lr=labels_lr, labels_full=labels_full)
        self._apply(sums, counts, hw, self.min_pixels, 0 if name == 'moving_average' else 1)
        return sums, counts
