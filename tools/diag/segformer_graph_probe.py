"""Losses of five warm-up steps of the mit_b1 SegFormer student: eager with the side streams, eager on one stream, HIP graph."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd.model.segformer import SegFormerStudent  # noqa: E402
from diga_amd.train_step import DigaTrainer  # noqa: E402
from oracle import mit as om, segformer_head as oh, synth  # noqa: E402

DEV = torch.device("cuda")


def make():
    m = SegFormerStudent("mit_b1")
    m.backbone.load_state_dict(om.state_dict(om.MIT_B1))
    m.backbone.reset_drop_path(0.0)
    m.final.load_state_dict(oh.state_dict())
    m.set_head_dropout(0.0)
    return m.to(DEV)


def run(graph):
    student, teacher = make(), make()
    teacher.train()
    tr = DigaTrainer(student, teacher, rng=random.Random(11), graph=graph)
    out = []
    for it in range(4):
        batch = [t.to(DEV) for t in synth.warmup_batch(900 + it, 2, 96, 128, block=16)]
        o = tr.warmup_step(it, *batch)
        out.append((round(float(o["ce"]), 6), round(float(o["distil"]), 6)))
    return out


print("eager, side streams:", run(False))
from diga_amd import config as _cfg
_cfg.DEFAULTS.teacher_stream = _cfg.DEFAULTS.wgrad_stream = False      # (round 6: fields, not environment variables)
print("eager, one stream  :", run(False))
print("graph              :", run(True))
