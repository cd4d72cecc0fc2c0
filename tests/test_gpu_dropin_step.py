"""A warm-up step driven through the DROP-IN surface a user of the reference keeps (the symbols
G5/train_DiGA_gta2city_warm_up.py:22-28 imports and the calls its loop :197-305 makes on them): `SegModel`, stock
`nn.Upsample(bilinear, align_corners=True)` to label size, full-resolution `cross_entropy2d` / `distillation_loss`,
`create/update_teacher_params`, a ClassMix written with plain torch ops, and stock `torch.optim.SGD` over
`student.optim_parameters(lr)` -- no DigaTrainer, no fused low-res loss block, no DigaSGD.  The caller is this test's own
code; the numbers it must reproduce are the capture of the reference.  Runs in a child process with
diga_amd/ in front of sys.path (the way a user of the reference would switch), three steps against the capture of the
reference (tests/golden/step.npz).  Stock SGD needs foreach=False: see INTEGRATION.md section 2."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import json, random, sys
import torch
# ---- the import surface a user of the reference keeps (module and symbol names of warm_up.py:22-28)
from model.model_noaux import SegModel
from util.loss import cross_entropy2d, distillation_loss
from util.utils import adjust_learning_rate, create_teacher_params, update_teacher_params
import model.model_noaux as _m
assert _m.__file__.startswith(sys.argv[1]), _m.__file__
sys.path.append(sys.argv[2])
from oracle import deeplab as od, detweights, synth           # test infrastructure: deterministic weights + inputs
from diga_amd import _lib
_lib.set_conv_math(int(sys.argv[3]))
foreach = {"0": False, "1": True, "none": None}[sys.argv[4]]

B, H, W = 2, 128, 128
BASE_LR, MAX_IT = 2.5e-4, 80000
dev = torch.device("cuda")
nets = {"stu": SegModel().to(dev), "tea": SegModel().to(dev)}
for net in nets.values():
    net.load_state_dict(detweights.state_dict(od.RESNET101))
    net.final.head[0].p = 0.0                                  # Dropout2d off, as in the capture
sgd = torch.optim.SGD(nets["stu"].optim_parameters(BASE_LR), lr=BASE_LR, momentum=0.9, weight_decay=5e-4,
                      **({} if foreach is None else {"foreach": foreach}))
to_label_size = torch.nn.Upsample(size=[H, W], mode="bilinear", align_corners=True)
nets["tea"] = create_teacher_params(nets["tea"], nets["stu"])
random.seed(77)
log = {"ce": [], "distil": [], "lr": []}


def classmix_mask(labels):
    # per image: half of the classes present (python `random`, the draw order of the capture), plus the ignore label
    m = torch.zeros_like(labels, dtype=torch.float32)
    for n in range(labels.shape[0]):
        present = torch.unique(labels[n]).tolist()
        chosen = set(random.sample(present, len(present) // 2)) | {255}
        m[n] = torch.isin(labels[n], torch.tensor(sorted(chosen), device=labels.device)).float()
    return m.unsqueeze(1)


for step in range(3):
    nets["stu"].train()
    adjust_learning_rate([sgd], base_lr=BASE_LR, i_iter=step, max_iter=MAX_IT, power=0.9)
    with torch.no_grad():
        nets["tea"] = update_teacher_params(nets["tea"], nets["stu"], step)
    img, img_aug, img_translated, lab = (t.to(dev) for t in synth.warmup_batch(1000 + step, B, H, W, block=16))
    m = classmix_mask(lab)
    mixed = img_translated * (1.0 - m) + img_aug * m           # selected classes come from the augmented source view
    both = torch.cat([img, mixed])
    stu_logits = to_label_size(nets["stu"](both)[2])
    tea_logits = to_label_size(nets["tea"](both)[2])
    ce = cross_entropy2d(stu_logits[:B], lab)
    kd = distillation_loss(tea_logits, stu_logits)
    sgd.zero_grad()
    (1.0 * ce + 0.5 * kd).backward()
    sgd.step()
    log["ce"].append(float(ce)); log["distil"].append(float(kd))
    log["lr"].append(sgd.param_groups[0]["lr"])
sd, td = nets["stu"].state_dict(), nets["tea"].state_dict()
log["student_head"] = sd["final.head.1.weight"].cpu().reshape(-1).tolist()
log["teacher_head"] = td["final.head.1.weight"].cpu().reshape(-1).tolist()
log["stem_sum"] = synth.checksum(sd["layer0.0.weight"].cpu())
log["stu_rm"] = sd["layer1.0.bn1.running_mean"].cpu().tolist()
nets["stu"].eval()
xp = synth.warmup_batch(2000, 1, H, W, block=16)[0].to(dev)
with torch.no_grad():
    log["probe_student"] = nets["stu"](xp)[2].cpu().reshape(-1).tolist()
print("DROPIN " + json.dumps(log))
"""


def _run(conv_math, foreach):
    env = dict(os.environ)
    pkg = os.path.join(ROOT, "diga_amd")
    env["PYTHONPATH"] = os.pathsep.join([pkg, ROOT])
    r = subprocess.run([sys.executable, "-c", SCRIPT, pkg, ROOT, str(conv_math), foreach], capture_output=True, text=True,
                       env=env, cwd="/tmp", timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DROPIN ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("conv_math", [0, 1], ids=["f32", "bf16x3"])
def test_dropin_surface_step_matches_reference_capture(golden, conv_math):
    import numpy as np
    g = golden("step")
    log = _run(conv_math, "0")
    for it in range(3):
        assert log["ce"][it] == pytest.approx(float(g["ce"][it]), rel=1e-3), it
        assert log["distil"][it] == pytest.approx(float(g["distil"][it]), rel=1e-3), it
        assert log["lr"][it] == pytest.approx(float(g["lr"][it]), rel=1e-12)
    for key, ref in (("student_head", g["student_head"]), ("teacher_head", g["teacher_head"])):
        got, want = np.asarray(log[key]), np.asarray(ref).reshape(-1)
        assert np.abs(got - want).max() <= 1e-5 + 5e-3 * np.abs(want).max(), key
    assert log["stem_sum"] == pytest.approx(float(g["ps_layer0_0_weight"]), rel=2e-3, abs=1e-3)
    got, want = np.asarray(log["stu_rm"]), np.asarray(g["stu_rm"])
    assert np.abs(got - want).max() <= 1e-5 + 1e-3 * np.abs(want).max()
    got, want = np.asarray(log["probe_student"]), np.asarray(g["probe_student"]).reshape(-1)
    assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max()


def test_stock_sgd_foreach_differs_on_duplicate_entries(golden):
    """What INTEGRATION.md section 2 warns about: torch 2.x's default (foreach=True on GPU tensors) walks the duplicate
    entries of `optim_parameters` differently from the single-tensor path of the torch 1.7.1 the reference pins -- the
    stem weight (multiplicity 2) ends up measurably elsewhere.  foreach=False (above) or DigaSGD reproduce the reference."""
    g = golden("step")
    log = _run(0, "none")
    ref = float(g["ps_layer0_0_weight"])
    # first-step losses do not depend on the optimizer; later parameters do
    assert log["ce"][0] == pytest.approx(float(g["ce"][0]), rel=1e-3)
    print("stem checksum: reference", ref, "stock SGD default foreach", log["stem_sum"])
