"""Drop-in import surface: with diga_amd/ in front on sys.path the reference scripts' own import lines
(G5/train_DiGA_gta2city_warm_up.py:22-28, ..._self_training.py:21-28) resolve to the build's modules."""
import os
import subprocess
import sys

from conftest import ROOT

SCRIPT = r"""
import sys
from model.model_noaux import SegModel, ImgEncoder, ImgDecoder
from util.loss import cross_entropy2d, distillation_loss
from util.metrics import runningScore
from util.utils import (adjust_learning_rate, save_models, load_models, create_teacher_params,
                        update_teacher_params, UnNormalize, Normalize, process_label)
from calc_centroids import Class_Features
import model.model_noaux as m, util.loss as l
assert m.__file__.startswith(sys.argv[1]) and l.__file__.startswith(sys.argv[1]), (m.__file__, l.__file__)
cf = Class_Features(numbers=19)
assert tuple(cf.objective_vectors.shape) == (19, 256) and cf.centroid_momentum == 1e-4
for name in ("get_centroid_weight", "get_centroid_distance", "calculate_mean_vector", "update_objective_SingleVector"):
    assert callable(getattr(cf, name))
import inspect
assert list(inspect.signature(cross_entropy2d).parameters) == ["input", "target", "weight", "size_average"]
assert list(inspect.signature(distillation_loss).parameters) == ["teacher_out", "student_out", "scale"]
assert list(inspect.signature(update_teacher_params).parameters) == ["teacher", "student", "iteration", "stage0", "mean", "replace"]
import torch
x = torch.arange(24, dtype=torch.float32).reshape(1, 1, 4, 6)
lab = torch.tensor([[[[0., 3.], [255., 18.]]]])
oh = process_label(lab, 19)
assert oh.shape == (1, 20, 2, 2) and float(oh[0, 19, 1, 0]) == 1.0 and float(oh[0, 3, 0, 1]) == 1.0
n = Normalize([0.5] * 1, [0.5] * 1)(x); u = UnNormalize([0.5] * 1, [0.5] * 1)(n)
assert torch.allclose(u, x)
print("dropin-ok")
"""


def test_reference_import_lines_resolve_to_the_build():
    env = dict(os.environ)
    pkg = os.path.join(ROOT, "diga_amd")
    env["PYTHONPATH"] = os.pathsep.join([pkg, ROOT])
    r = subprocess.run([sys.executable, "-c", SCRIPT, pkg], capture_output=True, text=True, env=env, cwd="/tmp",
                       timeout=300)
    assert r.returncode == 0 and "dropin-ok" in r.stdout, r.stderr[-2000:]
