"""One self-training step driven through the DROP-IN surface the way the unmodified reference script drives it
(G5/train_DiGA_gta2city_self_training.py:21-28 import lines, :214-387 loop body): `SegModel`, stock `nn.Upsample`,
full-resolution `cross_entropy2d` / `distillation_loss`, `update_teacher_params`, both inline ClassMix blocks with torch ops,
the bilateral-consensus block through `Class_Features.get_centroid_weight` (:298-304), the centroid updates through the
reference API one vector at a time (`calculate_mean_vector` -> `update_objective_SingleVector`, :327-341) and stock
`torch.optim.SGD(foreach=False)` -- no DigaTrainer, no fused loss block, no fused centroid pass -- against the capture of the
reference (tests/golden/selftrain.npz), in a child process with diga_amd/ in front of sys.path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import json, random, sys
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
# ---- the reference script's own import lines (self_training.py:21-28)
from model.model_noaux import SegModel
from util.loss import cross_entropy2d, distillation_loss
from util.utils import adjust_learning_rate, create_teacher_params, update_teacher_params
from calc_centroids import Class_Features
import calc_centroids as _c
assert _c.__file__.startswith(sys.argv[1]), _c.__file__
sys.path.append(sys.argv[2])
from oracle import deeplab as od, detweights, synth           # test infrastructure: deterministic weights + inputs
from diga_amd import _lib
_lib.set_conv_math(int(sys.argv[3]))

B, H, W = 2, 128, 128
student, teacher = SegModel().cuda(), SegModel().cuda()
for mdl in (student, teacher):
    mdl.load_state_dict(detweights.state_dict(od.RESNET101))
    mdl.final.head[0].p = 0.0
opt = optim.SGD(student.optim_parameters(2.5e-4), lr=2.5e-4, momentum=0.9, weight_decay=0.0005, foreach=False)
up = nn.Upsample(size=[H, W], mode='bilinear', align_corners=True)
teacher = create_teacher_params(teacher, student)
cf = Class_Features(numbers=19)
cents0 = torch.randn((19, 256), generator=synth.gen(7)) * 0.3
cf.objective_vectors = cents0.clone().cuda()
random.seed(78)
it = 3
student.train()
adjust_learning_rate([opt], base_lr=2.5e-4, i_iter=it, max_iter=80000, power=0.9)
with torch.no_grad():
    teacher = update_teacher_params(teacher, student, it)
x, x_aug, rec, lab, t_img, t_aug, pseudo_prob = (t.cuda() for t in synth.selftrain_batch(3000, B, H, W, block=16))
# ---- ClassMix #1 (self_training.py:259-275)
mask = torch.zeros(lab.size()).cuda()
for i in range(B):
    present = torch.unique(lab[i]).tolist()
    pick = random.sample(present, len(present) // 2)
    if 255 not in pick:
        pick.append(255)
    for c in pick:
        mask[i][lab[i] == c] = 1
mix = torch.zeros(rec.size()).cuda()
for i in range(B):
    mix[i] = torch.mul(rec[i], 1 - mask[i]) + torch.mul(x_aug[i], mask[i])
cat = torch.cat([x, mix])
_, _, s_cat, _ = student(cat)
with torch.no_grad():
    _, _, t_cat_lr, t_feat_cat = teacher(cat)
t_aug_raw = t_cat_lr[B:]
t_cat = up(t_cat_lr)
s_feat_tea_aug = t_feat_cat[B:]
# ---- bilateral consensus (:298-304)
with torch.no_grad():
    pseudo = pseudo_prob.clone()
    _, _, tt_pred, tt_feat = teacher(t_img)
    fw = up(cf.get_centroid_weight(tt_feat.detach()))
    feat_pseudo = fw.max(1, keepdim=True)[1].squeeze(1)
    pseudo[pseudo_prob != feat_pseudo] = 255
# ---- ClassMix #2 with label paste (:306-325)
cross_lab = pseudo.clone()
mask = torch.zeros(lab.size()).cuda()
for i in range(B):
    present = torch.unique(lab[i]).tolist()
    pick = random.sample(present, len(present) // 2)
    if 255 not in pick:
        pick.append(255)
    for c in pick:
        cross_lab[i][lab[i] == c] = c
        mask[i][lab[i] == c] = 1
cross_mix = torch.zeros(t_aug.size()).cuda()
for i in range(B):
    cross_mix[i] = torch.mul(t_aug[i], 1 - mask[i]) + torch.mul(x[i], mask[i])
cross_lab = cross_lab.long()
# ---- centroid updates through the reference API, one vector at a time (:327-341)
with torch.no_grad():
    nl_t = F.interpolate(pseudo.clone().reshape([B, 1, H, W]).float(), size=tt_feat.size()[2:], mode="nearest")
    v_t, id_t = cf.calculate_mean_vector(tt_feat, tt_pred.detach(), nl_t)
    for k in range(len(id_t)):
        cf.update_objective_SingleVector(id_t[k], v_t[k].detach(), start_mean=False)
    nl_s = F.interpolate(lab.clone().reshape([B, 1, H, W]).float(), size=s_feat_tea_aug.size()[2:], mode="nearest")
    v_s, id_s = cf.calculate_mean_vector(s_feat_tea_aug, t_aug_raw.detach(), nl_s)
    for k in range(len(id_s)):
        cf.update_objective_SingleVector(id_s[k], v_s[k].detach(), start_mean=False)
_, _, c_pred, _ = student(cross_mix)
c_pred = up(c_pred)
s_up = up(s_cat)
ce = cross_entropy2d(s_up[:B], lab)
di = distillation_loss(t_cat, s_up)
ce_mix = cross_entropy2d(c_pred, cross_lab)
total = 1.0 * (ce + ce_mix) + 0.25 * di
opt.zero_grad()
total.backward()
opt.step()
sd = student.state_dict()
print("DROPIN " + json.dumps({
    "ce": float(ce), "distil": float(di), "ce_mix": float(ce_mix), "total": float(total),
    "ids_t": [int(v) for v in id_t], "ids_s": [int(v) for v in id_s],
    "cents": cf.objective_vectors.cpu().reshape(-1).tolist(), "nums": cf.objective_vectors_num.cpu().tolist(),
    "pseudo_mismatch": int((pseudo.cpu() != torch.from_numpy(__import__("numpy").load(sys.argv[4])["pseudo"])).sum()),
    "cross_lab_mismatch": int((cross_lab.cpu() != torch.from_numpy(__import__("numpy").load(sys.argv[4])["cross_lab"])).sum()),
    "student_head": sd["final.head.1.weight"].cpu().reshape(-1).tolist(),
    "ps_layer3": synth.checksum(sd["layer3.10.conv2.weight"].cpu())}))
"""


@pytest.mark.parametrize("conv_math", [0, 1], ids=["f32", "bf16x3"])
def test_unmodified_selftraining_body_matches_reference_capture(golden, conv_math):
    g = golden("selftrain")
    env = dict(os.environ)
    pkg = os.path.join(ROOT, "diga_amd")
    env["PYTHONPATH"] = os.pathsep.join([pkg, ROOT])
    r = subprocess.run([sys.executable, "-c", SCRIPT, pkg, ROOT, str(conv_math), os.path.join(ROOT, "tests", "golden", "selftrain.npz")],
                       capture_output=True, text=True, env=env, cwd="/tmp", timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    log = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("DROPIN ")][-1][7:])
    for k in ("ce", "distil", "ce_mix", "total"):
        assert log[k] == pytest.approx(float(g[k]), rel=1e-3), k
    assert log["ids_t"] == g["ids_t"].tolist() and log["ids_s"] == g["ids_s"].tolist()
    # consensus labels: identical except where the upsampled centroid weights nearly tie (the capture stores the margins)
    near_tie = int((np.asarray(g["margin"]) < 1e-4).sum())
    assert log["pseudo_mismatch"] <= near_tie and log["cross_lab_mismatch"] <= near_tie
    cents = np.asarray(log["cents"]).reshape(19, 256)
    assert np.abs(cents - g["cents"]).max() <= 1e-6 + 1e-4 * np.abs(g["cents"] - g["cents0"]).max() + 2e-7 * np.abs(g["cents"]).max()
    assert log["nums"] == g["nums"].tolist()
    head, want = np.asarray(log["student_head"]), np.asarray(g["student_head"]).reshape(-1)
    assert np.abs(head - want).max() <= 1e-5 + 5e-3 * np.abs(want).max()
    assert log["ps_layer3"] == pytest.approx(float(g["ps_layer3"]), rel=2e-3, abs=1e-3)
