#!/usr/bin/env python3
"""Kernel-level timing of backward-weight: twin path (diga_conv2d_wgrad_twin, twins prebuilt) vs the register-staged
split-bf16 path (diga_conv2d_wgrad_nhwc_f32 in bf16x3 mode) on C2 layer shapes (16 images, 97x97)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402

SHAPES = [("l3.conv2", 256, 256, 3, 2), ("l4.conv2", 512, 512, 3, 4), ("aspp.d12", 2048, 256, 3, 12),
          ("aspp.bottleneck", 1280, 256, 3, 1), ("l3.conv1", 1024, 256, 1, 1), ("l3.conv3", 256, 1024, 1, 1)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    _lib.set_conv_math(1)
    dev, n, hw = "cuda", 16, 97
    m = n * hw * hw
    for name, cin, cout, k, dil in SHAPES:
        x = torch.randn((n, hw, hw, cin), device=dev)
        dy = torch.randn((n, hw, hw, cout), device=dev)
        pad = dil * (k // 2)
        xt = torch.empty(m * cin * 4, dtype=torch.uint8, device=dev)
        dyt = torch.empty(m * cout * 4, dtype=torch.uint8, device=dev)
        _lib.call("diga_make_twin", _lib.ptr(x), cin, _lib.ptr(xt), m, cin, _lib.stream())
        _lib.call("diga_make_twin", _lib.ptr(dy), cout, _lib.ptr(dyt), m, cout, _lib.stream())
        dw1 = torch.empty((cout, k, k, cin), device=dev)
        dw2 = torch.empty_like(dw1)
        ws1 = torch.empty(_lib.lib.diga_conv2d_wgrad_twin_workspace_bytes(n, hw, hw, cout, cin, k, k), dtype=torch.uint8, device=dev)
        ws2 = torch.empty(_lib.lib.diga_conv2d_wgrad_workspace_bytes(n, hw, hw, cout, cin, k, k), dtype=torch.uint8, device=dev)

        def twin():
            _lib.call("diga_conv2d_wgrad_twin", _lib.ptr(dyt), _lib.ptr(xt), _lib.ptr(dw1), _lib.ptr(ws1), ws1.numel(), n, hw, hw, cin,
                      hw, hw, cout, k, k, 1, 1, -pad, -pad, dil, dil, _lib.stream())

        def staged():
            _lib.call("diga_conv2d_wgrad_nhwc_f32", _lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw2), _lib.ptr(ws2), ws2.numel(), n, hw, hw, cin,
                      cin, hw, hw, cout, cout, k, k, 1, 1, -pad, -pad, dil, dil, 1, _lib.stream())

        t1, t2 = timed(twin), timed(staged)
        gf = 2.0 * m * cin * cout * k * k / 1e9
        err = float((dw1 - dw2).abs().max() / dw2.abs().max())
        print(f"{name:16s} twin {t1:7.3f} ms {gf / t1:7.1f} TF/s | staged {t2:7.3f} ms {gf / t2:7.1f} TF/s | rel diff {err:.1e}")
        del x, dy, xt, dyt, dw1, dw2, ws1, ws2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
