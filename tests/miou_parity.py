"""Fixed-seed validation-mIoU parity of the ResNet-101 build against the capture of the reference
(tests/golden/valmiou.npz, made by tools/gen_golden.py::gen_valmiou from G5/evaluate_val.py:73-93 + util/metrics.py:32-65):
the two-scale validation pass (diga_amd/evaluate.py: both forwards, fused upsample + max + argmax + confusion kernel)
on the same seeded synthetic val set and deterministic weights.

Used by tests/test_gpu_miou.py, and as a command (prints one JSON object) by bench.py's `miou_parity` leg:
    python tests/miou_parity.py
north_star bar: |mIoU - reference mIoU| <= 0.1 (percentage points)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def images(seed0, n_img, H, W):
    """The val images of gen_valmiou, regenerated from the seed (same generator calls, same order)."""
    from oracle import synth
    out = []
    for i in range(n_img):
        g = synth.gen(int(seed0) + i)
        out.append(torch.rand((1, 3, H, W), generator=g) * 2.0 - 1.0 + 0.5 * torch.randn((1, 3, 1, 1), generator=g))
    return out


def run(conv_math, dev="cuda"):
    from diga_amd import _lib
    from diga_amd.evaluate import evaluate_two_scale
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.util.metrics import runningScore
    from oracle import deeplab as od
    from oracle import detweights
    with np.load(os.path.join(ROOT, "tests", "golden", "valmiou.npz")) as z:
        g = {k: z[k] for k in z.files}
    n_img, H, W = (int(v) for v in g["geometry"])
    prev = _lib.get_conv_math()
    _lib.set_conv_math(conv_math)
    try:
        m = SegModel()
        m.load_state_dict(detweights.state_dict(od.RESNET101))
        m = m.to(dev).eval()
        rs = runningScore(19)
        agree = 0
        for i, img in enumerate(images(g["seed0"], n_img, H, W)):
            gt = torch.from_numpy(g["gt"][i:i + 1].astype(np.int64))
            pred = evaluate_two_scale(m, img.to(dev), gt.to(dev), rs, want_pred=True)
            agree += int((pred.cpu().numpy()[0] == g["pred"][i]).sum())
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            score, cls_iu = rs.get_scores()
    finally:
        _lib.set_conv_math(prev)
    miou = float(score["Mean IoU : \t"])
    return {"miou": miou, "miou_reference": float(g["miou"]), "miou_delta_points": 100.0 * abs(miou - float(g["miou"])),
            "pixels": n_img * H * W, "pixels_argmax_differs": n_img * H * W - agree,
            "confusion_abs_diff": float(np.abs(np.asarray(rs.confusion_matrix, dtype=np.float64) - g["hist"]).sum()),
            "iu_max_abs_diff": float(np.nanmax(np.abs(np.array([cls_iu[k] for k in range(19)]) - g["iu"])))}


if __name__ == "__main__":
    res = {name: run(math) for math, name in ((0, "f32"), (1, "bf16x3"))}
    res["workload"] = ("two-scale validation (full + half resolution, max of upsampled logits) of ResNet-101 with deterministic "
                       "weights on 3 seeded synthetic 256x512 images; reference = capture of G5/evaluate_val.py on PyTorch-CPU")
    print(json.dumps(res))
