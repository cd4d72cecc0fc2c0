"""GPU parity of the fused two-scale validation pass against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import deeplab as od
from oracle import detweights, synth
from oracle import evaluate as oe

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_two_scale_prediction_kernel_vs_oracle():
    from diga_amd import evaluate as ev
    from diga_amd.util.metrics import runningScore
    g = synth.gen(21)
    pred = 2.0 * torch.randn((2, 19, 17, 33), generator=g)
    pred_ds = 2.0 * torch.randn((2, 19, 9, 17), generator=g)
    gt = synth.block_labels(g, 2, 128, 256, block=16, ignore_frac=0.1)
    want, fused = oe.two_scale_prediction(pred, pred_ds, (128, 256))
    rs = runningScore(19, verbose=False)
    got = ev.two_scale_prediction(pred.to(DEV), pred_ds.to(DEV), (128, 256), gt.to(DEV), rs)
    top2 = fused.topk(2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert float(safe.float().mean()) > 0.999
    assert bool((got.cpu() == want)[safe].all())
    hist = rs.confusion_matrix
    assert hist.sum() == float((gt != 255).sum())
    want_hist = oe.confusion(gt.numpy(), got.cpu().numpy())         # same predictions -> identical matrix
    assert np.array_equal(hist, want_hist.astype(np.float64))


def test_evaluate_two_scale_tiny_model_vs_oracle():
    from diga_amd import evaluate as ev
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.util.metrics import runningScore
    sd = detweights.state_dict(od.TINY)
    m = SegModel(arch=sm.TINY)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    g = synth.gen(22)
    images = torch.rand((2, 3, 128, 192), generator=g) * 2 - 1
    labels = synth.block_labels(g, 2, 128, 192, block=16, ignore_frac=0.05)
    with torch.no_grad():
        want, want_hist, fused = oe.evaluate_two_scale(lambda x: od.forward(sd, x, od.TINY, training=False)[2], images, labels)
    rs = runningScore(19, verbose=False)
    got = ev.evaluate_two_scale(m, images.to(DEV), labels.to(DEV), rs, want_pred=True)
    top2 = fused.topk(2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * float(fused.abs().max())
    assert float(safe.float().mean()) > 0.95
    assert bool((got.cpu() == want)[safe].all())
    sc, _ = rs.get_scores()
    diff = np.abs(rs.confusion_matrix - want_hist).sum() / want_hist.sum()
    assert diff < 2.5 * float((~safe).float().mean()) + 1e-9
    assert 0.0 <= sc['Mean IoU : \t'] <= 1.0


def test_offline_passes_tiny_model():
    """Pseudo-label generation and the initial-centroid ('mean' mode) pass against the oracle."""
    from diga_amd import evaluate as ev
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    from oracle import centroids as oc
    sd = detweights.state_dict(od.TINY)
    m = SegModel(arch=sm.TINY)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    g = synth.gen(23)
    images = torch.rand((2, 3, 96, 128), generator=g) * 2 - 1
    pl = ev.generate_pseudo_labels(m, images.to(DEV))
    assert pl.dtype == torch.uint8 and tuple(pl.shape) == (2, 96, 128) and int(pl.max()) < 19
    with torch.no_grad():
        want, _, fused = oe.evaluate_two_scale(lambda x: od.forward(sd, x, od.TINY, training=False)[2], images,
                                               torch.zeros((2, 96, 128), dtype=torch.int64))
    top2 = fused.topk(2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * float(fused.abs().max())
    assert bool((pl.cpu().long() == want)[safe].all())
    cf = ev.initial_centroids(m, [images.to(DEV)], epochs=2)
    with torch.no_grad():
        _, _, out, feat = od.forward(sd, images, od.TINY, training=False)
    vecs, ids, _ = oc.class_mean_vectors(feat, out, None)
    c, n = torch.zeros(19, 256), torch.zeros(19)
    for _ in range(2):
        oc.centroid_mean_apply(c, n, vecs, ids)
    assert torch.equal(cf.objective_vectors_num.cpu(), n)
    assert float((cf.objective_vectors.cpu() - c).abs().max()) < 1e-3 * float(c.abs().max() + 1e-6)
