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


@pytest.mark.timeout(600)
def test_validation_pass_at_the_reference_geometry(golden, conv_math):
    """The offline pass at ITS geometry (G5/evaluate_val.py:60,73-93): ONE 1024 x 2048 image + its 512 x 1024 half through
    ResNet-101 in eval mode (running-statistics BatchNorm, N = 1: 33 153- and 8 385-row GEMMs -- other tile counts / kernel choices
    than any training shape), against tests/golden/valmiou_full.npz, a capture of the reference `SegModel.eval()` on the same
    image: low-resolution logits of both scales elementwise, the fused argmax at 1024 x 2048 (may differ only where the capture's
    top-2 margin is below 1e-3 of the logit scale, and on no more pixels than the capture holds within 2e-5 / 2e-4 of the scale --
    fp32 / split bf16: 137 and 1161 of 2 097 152 pixels lie within 1e-5 / 1e-4), the 19 x 19 confusion matrix (L1 distance <= twice
    the share of differing pixels: a moved pixel leaves one cell and enters another) and the mIoU (within 0.1 point: north_star)."""
    from diga_amd import evaluate as ev
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.util.metrics import runningScore
    g = golden("valmiou_full")
    H, W = (int(v) for v in g["geometry"])
    gen = synth.gen(int(g["seed"]))
    img = torch.rand((1, 3, H, W), generator=gen) * 2.0 - 1.0 + 0.5 * torch.randn((1, 3, 1, 1), generator=gen)
    m = SegModel(arch=sm.RESNET101)
    m.load_state_dict(detweights.state_dict(od.RESNET101))
    m = m.to(DEV).eval()
    x = img.to(DEV)
    scale = float(g["logit_scale"])
    tol = 1e-3 if conv_math == 1 else 1e-4                        # north_star: logits within 1e-3 relative
    with torch.no_grad():
        lo = m(x)[2]
        lo_ds = m(ev.resize_bilinear_ac(x, (H // 2, W // 2)))[2]
    assert tuple(lo.shape) == (1, 19, 129, 257) and tuple(lo_ds.shape) == (1, 19, 65, 129)
    e1 = float((lo[0].cpu() - g.t("logits")).abs().max()) / scale
    e2 = float((lo_ds[0].cpu() - g.t("logits_ds")).abs().max()) / scale
    rs = runningScore(19, verbose=False)
    gt = g.t("gt").long()[None]
    pred = ev.evaluate_two_scale(m, x, gt.to(DEV), rs, want_pred=True)[0].cpu()
    want = g.t("pred").long()
    differs = pred != want
    near = torch.from_numpy(np.unpackbits(g["near_tie_bits"])[: H * W].reshape(H, W).astype(bool))
    share = float(differs.float().mean())
    sc, _ = rs.get_scores()
    hist_l1 = float(np.abs(rs.confusion_matrix - g["hist"]).sum() / g["hist"].sum())
    miou = 100.0 * float(sc["Mean IoU : \t"])
    print(f"\n[valmiou_full / math {conv_math}] logits {e1:.2e} / {e2:.2e} of scale; argmax differs on {int(differs.sum())} of {H * W} pixels "
          f"({share:.2e}; capture: {g['near_ties'].tolist()} within 1e-5/1e-4/1e-3); confusion L1 {hist_l1:.2e}; "
          f"mIoU {miou:.4f} vs {100 * float(g['miou']):.4f}")
    assert e1 < tol and e2 < tol, (e1, e2)
    assert not bool((differs & ~near).any()), int((differs & ~near).sum())
    allowed = 2 * int(g["near_ties"][0 if conv_math == 0 else 1])           # pixels within 1e-5 (fp32) / 1e-4 (split bf16) of a tie, x 2
    assert int(differs.sum()) <= allowed, (int(differs.sum()), allowed)
    assert hist_l1 <= 2.0 * share + 1e-12, (hist_l1, share)
    assert abs(miou - 100 * float(g["miou"])) <= 0.1
