#!/usr/bin/env python3
"""Which host-side ops launch the small device copies of a warm-up step?  One profiled C2 step under torch.profiler,
grouped by (op name, python call site).  Diagnostic only.

    python tools/find_copies.py
"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diga_amd import _lib, synthetic  # noqa: E402
from diga_amd.model import seg_model_noaux as sm  # noqa: E402
from diga_amd.model.model_noaux import SegModel  # noqa: E402
from diga_amd.train_step import DigaTrainer  # noqa: E402


def main():
    dev = "cuda"
    _lib.set_conv_math(1)
    torch.manual_seed(0)
    student, teacher = SegModel(arch=sm.RESNET101).to(dev), SegModel(arch=sm.RESNET101).to(dev)
    teacher.train()
    tr = DigaTrainer(student, teacher, rng=random.Random(1))
    batch = synthetic.warmup_batch(1, 8, 768, 768, block=64, device=dev)
    for i in range(2):
        tr.warmup_step(i, *batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.warmup_step(2, *batch)
        torch.cuda.synchronize()
    rows = []
    for ev in prof.key_averages(group_by_stack_n=6):
        if any(k in ev.key for k in ("copy", "Memcpy", "clone", "contiguous", "fill", "zero", "add", "mul", "sum", "cat", "stack")):
            rows.append((ev.count, ev.key, [s for s in ev.stack if "diga_amd" in s or "bench" in s][:2]))
    rows.sort(key=lambda r: -r[0])
    for c, k, st in rows[:40]:
        print(f"{c:6d}  {k:40s} {' <- '.join(s.strip()[-90:] for s in st)}")


if __name__ == "__main__":
    main()
