#!/usr/bin/env python3
"""41 warm-up steps of the C2 configuration; prints allocated / reserved / peak device memory and the loss at steps 3, 10,
20 and 40 -- a run-to-run stability check (no growth between steps; r02, split bf16: 1.6 GB live between steps, 72 GB peak,
115 GB reserved of 288 GB).  `--math f32` runs the headline arithmetic (LDS-DMA + Winograd kernels: the transformed inputs of
the 3x3 layers stay alive between forward and backward).

    python tools/leak_check.py [--math f32|bf16x3] [--steps 41]
"""
import os, sys, random, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from diga_amd import _lib, synthetic
from diga_amd.model import seg_model_noaux as sm
from diga_amd.model.model_noaux import SegModel
from diga_amd.train_step import DigaTrainer
import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--math', default='bf16x3', choices=['f32', 'bf16x3'])
ap.add_argument('--steps', type=int, default=41)
A = ap.parse_args()
_lib.set_conv_math(1 if A.math == 'bf16x3' else 0)
torch.manual_seed(0)
dev = "cuda"
student, teacher = SegModel(arch=sm.RESNET101).to(dev), SegModel(arch=sm.RESNET101).to(dev)
teacher.train()
tr = DigaTrainer(student, teacher, rng=random.Random(1))
batch = synthetic.warmup_batch(1, 8, 768, 768, block=64, device=dev)
for i in range(A.steps):
    out = tr.warmup_step(i, *batch)
    if i in (3, 10, 20, A.steps - 1):
        torch.cuda.synchronize()
        print(i, "alloc GB", round(torch.cuda.memory_allocated() / 2**30, 2), "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2),
              "max GB", round(torch.cuda.max_memory_allocated() / 2**30, 2), "loss", float(out["total"]))
