"""Oracle (CPU, test infrastructure only): centroid bank and dynamic pseudo-label selection.

Restates
  * G5/calc_centroids.py:166-176  feat_centroid_distance / get_centroid_weight
  * G5/train_DiGA_gta2city_self_training.py:298-304  bilateral consensus
  * G5/calc_centroids.py:120-145  calculate_mean_vector (+ G5/util/utils.py:158-163 process_label)
  * G5/calc_centroids.py:147-164  update_objective_SingleVector
SURVEY App. A-6/A-7: the distance is L2 (not cosine); argmax is taken AFTER the
bilinear upsampling of the 19 weight maps; the centroid EMA is applied
sequentially in image-major / class-minor order.
Pinned by tests/golden (G-centroid, G-meanvec, KAT-3, KAT-5).
"""
import torch
import torch.nn.functional as F

from .losses import upsample_bilinear_ac

IGNORE = 255


def centroid_distance(feat, centroids):
    """d[n,i,y,x] = || centroids[i] - feat[n,:,y,x] ||_2   (calc_centroids.py:166-171)."""
    diff = centroids[None, :, :, None, None] - feat[:, None]       # [N,K,C,h,w]
    return diff.pow(2).sum(2).sqrt()


def centroid_weight(feat, centroids):
    """softmax(-d) over the class axis (calc_centroids.py:173-176)."""
    return F.softmax(-centroid_distance(feat, centroids), dim=1)


def consensus_filter(weights_lr, pseudo_prob):
    """self_training.py:301-304: upsample weights to label size, argmax over
    classes (first max wins), keep the offline pseudo-label only where it agrees.
    Returns (filtered_pseudo, feat_pseudo)."""
    w = upsample_bilinear_ac(weights_lr, pseudo_prob.shape[-2:])
    feat_pseudo = w.argmax(dim=1)
    out = pseudo_prob.clone()
    out[pseudo_prob != feat_pseudo] = IGNORE
    return out, feat_pseudo


def nearest_downsample_labels(labels, size):
    """F.interpolate(labels.float()[:,None], size, mode='nearest'): src = floor(dst*in/out)
    (self_training.py:328-330)."""
    H, W = labels.shape[-2:]
    h, w = size
    ys = (torch.arange(h, dtype=torch.float32) * (H / h)).floor().to(torch.int64).clamp_(max=H - 1)
    xs = (torch.arange(w, dtype=torch.float32) * (W / w)).floor().to(torch.int64).clamp_(max=W - 1)
    return labels[:, ys][:, :, xs]


def class_ids(outputs, labels_lr=None, n_classes=19):
    """Per low-res pixel: the class whose one-hot survives
    onehot20(argmax softmax(out)) * onehot20(label) (calc_centroids.py:121-128),
    or n_classes when nothing survives.  labels_lr is [N,h,w] (any numeric dtype)
    with values >= n_classes mapped to the dead 20th bucket (utils.py:160)."""
    pred = outputs.argmax(dim=1)             # softmax is monotone: same argmax
    if labels_lr is None:
        return pred
    lab = labels_lr.to(torch.int64)
    dead = torch.full_like(pred, n_classes)
    return torch.where((lab == pred) & (lab < n_classes), pred, dead)


def class_mean_vectors(feat, outputs, labels_lr=None, n_classes=19, min_pixels=5):
    """calculate_mean_vector (calc_centroids.py:120-145).  Returns (vectors, ids,
    image_index): per (image n, class t) in n-major / t-minor order, for classes
    with at least `min_pixels` member pixels, the mean feature vector [C]."""
    ids_map = class_ids(outputs, labels_lr, n_classes)
    N, C, h, w = feat.shape
    vectors, ids, owners = [], [], []
    for n in range(N):
        for t in range(n_classes):
            m = (ids_map[n] == t)
            cnt = int(m.sum())
            if cnt == 0 or cnt < min_pixels:
                continue
            mf = m.to(feat.dtype)
            scale = mf.mean()                               # adaptive_avg_pool2d(mask,1)
            v = (feat[n] * mf).mean(dim=(1, 2)) / scale     # avg_pool(feat*mask)/scale
            vectors.append(v)
            ids.append(t)
            owners.append(n)
    return vectors, ids, owners


def centroid_ema_apply(centroids, nums, vectors, ids, momentum=1e-4, cap=3000.0):
    """update_objective_SingleVector(name='moving_average', start_mean=False)
    applied sequentially (calc_centroids.py:147-156).  In-place on centroids/nums."""
    for v, t in zip(vectors, ids):
        if float(v.sum()) == 0.0:
            continue
        centroids[t] = centroids[t] * (1.0 - momentum) + momentum * v
        nums[t] = min(float(nums[t]) + 1.0, cap)
    return centroids, nums


def centroid_mean_apply(centroids, nums, vectors, ids, cap=3000.0):
    """update_objective_SingleVector(name='mean') (calc_centroids.py:157-162):
    running mean used by the offline initial-centroid pass."""
    for v, t in zip(vectors, ids):
        if float(v.sum()) == 0.0:
            continue
        centroids[t] = centroids[t] * nums[t] + v
        nums[t] = nums[t] + 1.0
        centroids[t] = centroids[t] / nums[t]
        nums[t] = min(float(nums[t]), cap)
    return centroids, nums
