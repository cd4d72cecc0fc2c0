#!/usr/bin/env python3
"""Where do the step's device-to-device memcpys (rocprof: __amd_rocclr_copyBuffer, ~365 per C2 step) come from?  One warm-up step under
torch.profiler with stacks; prints the Python call sites of every op that launched a Memcpy."""
import collections
import os
import random
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from diga_amd import _lib, synthetic, train_step
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    dev = torch.device("cuda", 0)
    _lib.set_conv_math(0)
    if os.environ.get("FC_CONFIG") == "c5":                                  # SegFormer-B5 student / teacher (BASELINE configs[4])
        from diga_amd.model.segformer import SegFormerStudent
        torch.manual_seed(0)
        student, teacher = SegFormerStudent("mit_b5", head="segformer").to(dev), SegFormerStudent("mit_b5", head="segformer").to(dev)
    else:
        student, teacher = SegModel(arch=sm.RESNET101).to(dev), SegModel(arch=sm.RESNET101).to(dev)
    teacher.train()
    tr = train_step.DigaTrainer(student, teacher, rng=random.Random(1))
    batch = synthetic.warmup_batch(1234, int(os.environ.get("FC_B", "2")), int(os.environ.get("FC_HW", "256")), int(os.environ.get("FC_HW", "256")), block=32, device=dev)
    for it in range(2):
        tr.warmup_step(it, *batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.warmup_step(2, *batch)
        torch.cuda.synchronize()
    sites = collections.Counter()
    names = collections.Counter()
    for ev in prof.events():
        if "emcpy" in ev.name or "copyBuffer" in ev.name:
            names[ev.name] += 1
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::cat", "aten::fill_", "aten::zero_",
                       "aten::add_", "aten::add", "aten::sum", "aten::mul", "aten::zeros", "aten::zeros_like", "aten::_foreach_add_",
                       "aten::_foreach_mul_", "aten::reshape", "aten::detach"):
            if ev.name in ("aten::reshape", "aten::detach", "aten::to", "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::clone"):
                continue                                   # wrappers: the copy_ / fill_ under them is counted
            st = [s for s in (ev.stack or []) if "diga_amd" in s or "bench.py" in s]
            where = st[0] if st else ("autograd engine" if any("run_backward" in s or "autograd" in s for s in (ev.stack or [])) else "?")
            sites[(ev.name, where)] += 1
    print("memcpy-like events:", dict(names))
    for (n, s), c in sites.most_common(40):
        print(f"{c:5d}  {n:18s} {s}")


if __name__ == "__main__":
    main()
