#!/usr/bin/env python3
"""K-loop efficiency of the fp32 LDS-DMA conv kernels without tile-count quantisation: shapes whose tile count is a whole
number of rounds of 256 blocks (M = 2 x 256 x 256 pixels, Cout = 256 -> 1024 tiles = 4 rounds), long and short K."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402
from diga_amd.model.conv import DigaConv2d  # noqa: E402

_lib.set_conv_math(0)
PEAK = 157.3


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for name, n, cin, h, cout, k, d in [("3x3 K=18432", 2, 2048, 256, 256, 3, 1), ("1x1 K=2048", 2, 2048, 256, 256, 1, 1),
                                     ("1x1 K=1024", 2, 1024, 256, 256, 1, 1), ("1x1 K=256 ->1024", 2, 256, 256, 1024, 1, 1),
                                     ("3x3 K=2304", 2, 256, 256, 256, 3, 2)]:
    m = DigaConv2d(cin, cout, k, padding=d * (k // 2), dilation=d, bias=False).cuda()
    x = torch.randn(n, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
    y = m(x)
    g = torch.randn_like(y)
    fl = 2.0 * n * h * h * cout * cin * k * k
    t_f = timed(lambda: m(x))
    def bwd():
        x.grad = None
        m.weight.grad = None
        y2 = m(x)
        y2.backward(g)
    t_fb = timed(bwd)
    print(f"{name:20s} fwd {t_f:8.3f} ms {fl / t_f / 1e9:7.1f} TF/s {fl / t_f / 1e9 / PEAK:5.3f} | fwd+dgrad+wgrad {t_fb:8.3f} ms "
          f"{3 * fl / t_fb / 1e9:7.1f} TF/s {3 * fl / t_fb / 1e9 / PEAK:5.3f}")
