#!/usr/bin/env python3
"""Which stream form of the self-training step changes under config.wgrad_hold?  Two steps at the golden's geometry per (overlap, hold),
twice each: losses of the second step and a checksum of the student after it."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from diga_amd import config  # noqa: E402
from diga_amd import train_step as ts  # noqa: E402
from diga_amd.calc_centroids import Class_Features  # noqa: E402
from diga_amd.model.model_noaux import SegModel  # noqa: E402
from oracle import deeplab as od  # noqa: E402
from oracle import detweights, synth  # noqa: E402


def run(overlap, hold, wgrad=True):
    cfg = config.DEFAULTS.replace(c4_overlap=overlap, wgrad_hold=hold, wgrad_stream=wgrad)

    def make():
        m = SegModel()
        m.load_state_dict(detweights.state_dict(od.RESNET101))
        m.final.head[0].p = 0.0
        return m.to("cuda")
    student, teacher = make(), make()
    teacher.train()
    tr = ts.DigaTrainer(student, teacher, rng=random, config=cfg)
    cf = Class_Features(numbers=19)
    cf.objective_vectors = torch.randn((19, 256), generator=torch.Generator().manual_seed(5)).to("cuda")
    logs = []
    for it in (3, 4):
        batch = [t.to("cuda") for t in synth.selftrain_batch(3000 + it, 2, 128, 128, block=16)]
        random.seed(78 + it)
        logs.append({k: float(v) for k, v in tr.selftrain_step(it, *batch, cf).items()})
    torch.cuda.synchronize()
    cs = sum(float(v.double().abs().sum()) for v in student.state_dict().values() if v.is_floating_point())
    return logs[1]["total"], logs[1]["ce_mix"], cs


for overlap in (0, 1, 2):
    for hold, wg in ((0, True), (4, True), (1, True), (1000, True)):
        for rep in range(2):
            print(overlap, "hold", hold, "wgrad_stream", wg, "rep", rep, run(overlap, hold, wg), flush=True)
