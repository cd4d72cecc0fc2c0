#!/usr/bin/env python3
"""Diagnostic: per-layer weight-gradient deviations of the HIP path from the reference captures at the benchmark
geometries (tests/golden/full768.npz, full512x1024.npz), both conv arithmetics.  python tools/diag/fullsize_grad_errors.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402
from diga_amd.model import seg_model_noaux as sm  # noqa: E402
from diga_amd.model.model_noaux import SegModel  # noqa: E402
from oracle import deeplab as od, detweights, synth  # noqa: E402

for name in ("full512x1024", "full768"):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    _, H, W = (int(v) for v in g["geometry"])
    for math in (0, 1):
        _lib.set_conv_math(math)
        gen = synth.gen(int(g["seed"]))
        x = torch.rand((2, 3, H, W), generator=gen) * 2 - 1
        m = SegModel(arch=sm.RESNET101)
        m.load_state_dict(detweights.state_dict(od.RESNET101))
        m = m.cuda().train()
        m.final.head[0].p = 0.0
        out = m(x.cuda())[2]
        want = torch.from_numpy(g["out"])
        probe = torch.randn(want.shape, generator=gen)
        (out * probe.cuda()).sum().backward()
        named = {n.replace(".", "_"): p for n, p in m.named_parameters()}
        print(f"== {name} math={math} logits err {float((out.detach().cpu() - want).abs().max() / want.abs().max()):.2e}")
        for k in sorted(k[2:-5] for k in g if k.startswith("g_") and k.endswith("__sum")):
            gr = named[k].grad.detach().cpu()
            _, l1, l2 = (float(v) for v in g["g_" + k + "__sum"])
            step = int(g["g_" + k + "__step"])
            smp = torch.from_numpy(g["g_" + k + "__sample"])
            d = gr.reshape(-1)[::step] - smp
            print(f"{k:34s} L1 {abs(float(gr.abs().sum()) / l1 - 1):.1e} L2 {abs(float(gr.norm()) / l2 - 1):.1e} "
                  f"sample max {float(d.abs().max() / smp.abs().max()):.1e} relL2 {float(d.norm() / smp.norm()):.1e}")
        del m
        torch.cuda.empty_cache()
