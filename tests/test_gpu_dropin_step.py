"""A warm-up step driven through the DROP-IN surface exactly as the unmodified reference script drives it
(G5/train_DiGA_gta2city_warm_up.py:22-28 import lines, :145-185 set-up, :197-305 loop body): `SegModel`, stock
`nn.Upsample(bilinear, align_corners=True)` to label size, full-resolution `cross_entropy2d` / `distillation_loss`,
`create/update_teacher_params`, the inline ClassMix block with torch ops, and stock `torch.optim.SGD` over
`student.optim_parameters(lr)` -- no DigaTrainer, no fused low-res loss block, no DigaSGD.  Runs in a child process with
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
import torch.nn as nn
import torch.optim as optim
# ---- the reference script's own import lines (warm_up.py:22-28)
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

batch_size, H, W = 2, 128, 128
learning_rate_seg, num_steps, power = 2.5e-4, 80000, 0.9
lambda_seg, lambda_distil = 1.0, 0.5
student, teacher = SegModel().cuda(), SegModel().cuda()
for mdl in (student, teacher):
    mdl.load_state_dict(detweights.state_dict(od.RESNET101))
    mdl.final.head[0].p = 0.0                                  # Dropout2d off, as in the capture
kw = {} if foreach is None else {"foreach": foreach}
student_opt = optim.SGD(student.optim_parameters(learning_rate_seg), lr=learning_rate_seg, momentum=0.9,
                        weight_decay=0.0005, **kw)
seg_opt_list = [student_opt]
seg_loss = cross_entropy2d
upsample_src = nn.Upsample(size=[H, W], mode='bilinear', align_corners=True)
teacher = create_teacher_params(teacher, student)
random.seed(77)
log = {"ce": [], "distil": [], "lr": []}
for i_iter in range(3):
    student.train()
    adjust_learning_rate(seg_opt_list, base_lr=learning_rate_seg, i_iter=i_iter, max_iter=num_steps, power=power)
    with torch.no_grad():
        teacher = update_teacher_params(teacher, student, i_iter)
    sdatav, sdatav_aug, rec_s2t, slabelv = (t.cuda() for t in synth.warmup_batch(1000 + i_iter, batch_size, H, W, block=16))
    # Cross-domain Mixture Data Augmentation (warm_up.py:240-259, verbatim semantics)
    rec_s2t_clone = rec_s2t.detach().clone()
    sdatav_aug_clone = sdatav_aug.detach().clone()
    mask = torch.zeros(slabelv.size()).cuda()
    for idx in range(slabelv.size()[0]):
        label_list = torch.unique(slabelv[idx]).tolist()
        classes_select = random.sample(label_list, len(label_list) // 2)
        if 255 not in classes_select:
            classes_select.append(255)
        for cls_m in classes_select:
            mask[idx][slabelv[idx] == cls_m] = 1
    sdatav_aug_crdomix = torch.zeros(rec_s2t_clone.size()).cuda()
    for idx in range(rec_s2t_clone.size()[0]):
        sdatav_aug_crdomix[idx] = torch.mul(rec_s2t_clone[idx], 1 - mask[idx]) + torch.mul(sdatav_aug_clone[idx], mask[idx])
    sdatav_cat = torch.cat([sdatav, sdatav_aug_crdomix])
    _, _, s_pred_cat_stu, s_feat_cat_stu = student(sdatav_cat)
    s_pred_cat_stu = upsample_src(s_pred_cat_stu)
    s_pred_stu = s_pred_cat_stu[:batch_size]
    _, _, s_pred_cat_tea, s_feat_cat_tea = teacher(sdatav_cat)
    s_pred_cat_tea = upsample_src(s_pred_cat_tea)
    loss_semseg = seg_loss(s_pred_stu, slabelv)
    loss_s_distil = distillation_loss(s_pred_cat_tea, s_pred_cat_stu)
    total_loss = lambda_seg * loss_semseg + lambda_distil * loss_s_distil
    student_opt.zero_grad()
    total_loss.backward()
    student_opt.step()
    log["ce"].append(float(loss_semseg)); log["distil"].append(float(loss_s_distil))
    log["lr"].append(student_opt.param_groups[0]["lr"])
sd, td = student.state_dict(), teacher.state_dict()
log["student_head"] = sd["final.head.1.weight"].cpu().reshape(-1).tolist()
log["teacher_head"] = td["final.head.1.weight"].cpu().reshape(-1).tolist()
log["stem_sum"] = synth.checksum(sd["layer0.0.weight"].cpu())
log["stu_rm"] = sd["layer1.0.bn1.running_mean"].cpu().tolist()
student.eval()
xp = synth.warmup_batch(2000, 1, H, W, block=16)[0].cuda()
with torch.no_grad():
    log["probe_student"] = student(xp)[2].cpu().reshape(-1).tolist()
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
def test_unmodified_script_body_matches_reference_capture(golden, conv_math):
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
