"""Two-scale validation pass of the DiGA scripts on MI355X:
G5/train_DiGA_gta2city_warm_up.py:343-373 and G5/evaluate_val.py:73-93 -- the model sees the image at full and
at half resolution, both logit maps are upsampled (bilinear, align_corners) to label size, their element-wise
maximum is arg-maxed and scored with `runningScore`.

The upsample / max / argmax / confusion chain runs in one kernel (`diga_two_scale_confusion`): the
[N,19,1024,2048] upsampled tensors of the reference (159 MB each, per image) are never materialised and the
prediction never leaves the device.
"""
import torch

from diga_amd import _lib


def resize_bilinear_ac(x, size):
    """nn.functional.interpolate(x, size, mode='bilinear', align_corners=True) for NCHW fp32."""
    _lib.require_gpu(x)
    x = _lib.contiguous(x, torch.float32)
    n, c, h, w = x.shape
    out = torch.empty((n, c, size[0], size[1]), dtype=torch.float32, device=x.device)
    _lib.call("diga_upsample_bilinear_ac", _lib.ptr(x), _lib.ptr(out), n * c, h, w, size[0], size[1], _lib.stream())
    return out


def two_scale_prediction(pred, pred_ds, size, gt=None, running=None, want_pred=True):
    """argmax_k max(up(pred), up(pred_ds)) at `size`; when `gt` and a diga_amd.util.metrics.runningScore are
    given, its confusion matrix is updated on the device."""
    _lib.require_gpu(pred, pred_ds)
    a = _lib.contiguous(pred.detach(), torch.float32)
    b = _lib.contiguous(pred_ds.detach(), torch.float32)
    n, k, ha, wa = a.shape
    _, _, hb, wb = b.shape
    H, W = size
    out = torch.empty((n, H, W), dtype=torch.int64, device=a.device) if want_pred else None
    g = hist = None
    if gt is not None and running is not None:
        g = _lib.contiguous(gt.to(a.device), torch.int64)
        hist = running._dev_hist()
    _lib.call("diga_two_scale_confusion", _lib.ptr(a), ha, wa, _lib.ptr(b), hb, wb, _lib.ptr(g), _lib.ptr(out),
              _lib.ptr(hist), n, k, H, W, _lib.stream())
    return out


@torch.no_grad()
def evaluate_two_scale(model, images, labels, running, ds_size=None, want_pred=False):
    """One validation batch: images [N,3,H,W], labels [N,H,W].  The model must be in eval() mode (the caller
    decides, as the reference scripts do)."""
    H, W = labels.shape[-2:]
    ds_size = ds_size or (images.shape[-2] // 2, images.shape[-1] // 2)
    image_ds = resize_bilinear_ac(images, ds_size)
    pred = model(images)[2]
    pred_ds = model(image_ds)[2]
    return two_scale_prediction(pred, pred_ds, (H, W), labels, running, want_pred)


@torch.no_grad()
def generate_pseudo_labels(model, images, size=None, ds_size=None):
    """Offline pseudo-label pass (G5/pseudolabel_generator.py:69-86): argmax of the softmax of the two-scale
    max-logits (softmax is monotone, so the argmax is taken on the fused logits directly) as uint8 train ids,
    ready to be written as palette PNGs by the caller."""
    size = size or tuple(images.shape[-2:])
    ds_size = ds_size or (images.shape[-2] // 2, images.shape[-1] // 2)
    pred = model(images)[2]
    pred_ds = model(resize_bilinear_ac(images, ds_size))[2]
    return two_scale_prediction(pred, pred_ds, size).to(torch.uint8)


@torch.no_grad()
def initial_centroids(model, target_batches, class_features=None, epochs=5):
    """Initial class centroids on the target domain (G5/calc_centroids.py:17-81, target branch): per batch the
    class-mean feature vectors of the model's own predictions update the bank in 'mean' mode.  `target_batches`
    is a re-iterable of image tensors; returns the Class_Features (its objective_vectors is what the reference
    torch.save()s to <centroid_dir>/feat_centroids)."""
    from diga_amd.calc_centroids import Class_Features
    cf = class_features or Class_Features(numbers=19)
    for _ in range(epochs):
        for images in target_batches:
            _, _, out, feat = model(images)
            cf.update_from_batch(feat, out, name='mean')
    return cf
