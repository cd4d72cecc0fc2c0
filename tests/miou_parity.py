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


def train_and_validate(g, seed, conv_math, dev="cuda"):
    """One seed of the TRAINED-model experiment of tests/golden/trainmiou.npz (tools/gen_golden.py::gen_trainmiou: the reference's
    warm-up loop for 300 steps on the learnable synthetic task, Dropout2d live, its two-scale validation on 32 held-out images every
    50 steps) on the HIP path: same data seeds, same ClassMix draws, the device's own Dropout2d stream.  Returns (mIoU curve, final
    mIoU, mean CE of the last `every` steps)."""
    import random
    from diga_amd import _lib
    from diga_amd import evaluate as ev
    from diga_amd.model.model_noaux import SegModel
    from diga_amd.train_step import DigaTrainer
    from diga_amd.util.metrics import runningScore
    from oracle import deeplab as od
    from oracle import detweights, synth
    B, H, W, steps, block, every, n_val, data_seed0, val_seed0 = (int(v) for v in g["geometry"])

    def model():
        m = SegModel()
        m.load_state_dict(detweights.state_dict(od.RESNET101))
        return m.to(dev)

    def validate(student):
        rs = runningScore(19, verbose=False)
        student.eval()
        for i in range(n_val):
            img, _, _, gt = synth.learnable_batch(val_seed0 + i, 1, H, W, block=block)
            ev.evaluate_two_scale(student, img.to(dev), gt.to(dev), rs)
        student.train()
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            sc, _ = rs.get_scores()
        return float(sc["Mean IoU : \t"])

    prev = _lib.get_conv_math()
    _lib.set_conv_math(conv_math)
    try:
        torch.manual_seed(1234 + seed)                            # the device's Dropout2d stream (the reference drew from the CPU generator)
        random.seed(4321 + seed)                                  # ClassMix: the same draws as the capture
        student, teacher = model(), model()                       # Dropout2d(0.1) LIVE in both heads, as in the reference
        teacher.train()
        tr = DigaTrainer(student, teacher, base_lr=float(g["lr"]), rng=random)
        curve, ce = [], []
        for it in range(steps):
            batch = synth.learnable_batch(data_seed0 + 1000 * seed + it, B, H, W, block=block)
            log = tr.warmup_step(it, *(t.to(dev) for t in batch))
            ce.append(log["ce"])
            if (it + 1) % every == 0:
                curve.append(validate(student))
        final = validate(student)
        return np.array(curve), final, float(torch.stack(ce[-every:]).mean())
    finally:
        _lib.set_conv_math(prev)
        _lib.join_side()


def trained_model(seed=0, conv_math=0):
    """bench.py's `miou_parity.trained_model`: ONE seed of the experiment run live (~40 s) next to the capture's three reference runs."""
    with np.load(os.path.join(ROOT, "tests", "golden", "trainmiou.npz")) as z:
        g = {k: z[k] for k in z.files}
    curve, final, tail = train_and_validate(g, seed, conv_math)
    ref = 100.0 * np.asarray(g["miou"], dtype=np.float64)
    seeds = [int(v) for v in g["seeds"]]
    return {"workload": "reference warm-up loop, 300 steps, B = 2 x 128 x 128 learnable synthetic task, Dropout2d live; two-scale validation on 32 "
                        "held-out images (tools/gen_golden.py::gen_trainmiou)",
            "seed": seed, "arithmetic": "f32" if conv_math == 0 else "bf16x3", "hip_final_miou": 100.0 * final,
            "hip_miou_curve": [100.0 * float(v) for v in curve], "reference_same_seed": float(ref[seeds.index(seed)]) if seed in seeds else None,
            "reference_final_miou_per_seed": ref.tolist(), "reference_mean": float(ref.mean()), "reference_seed_spread": float(ref.max() - ref.min()),
            "reference_curve_mean": (100.0 * np.asarray(g["curve"]).mean(axis=0)).tolist(),
            "delta_to_reference_mean_points": 100.0 * final - float(ref.mean()), "all_seeds_and_both_arithmetics": "tests/test_gpu_trainmiou.py"}


if __name__ == "__main__":
    res = {name: run(math) for math, name in ((0, "f32"), (1, "bf16x3"))}
    if "--trained" in sys.argv:
        res["trained_model"] = trained_model()
    res["workload"] = ("two-scale validation (full + half resolution, max of upsampled logits) of ResNet-101 with deterministic "
                       "weights on 3 seeded synthetic 256x512 images; reference = capture of G5/evaluate_val.py on PyTorch-CPU")
    print(json.dumps(res))
