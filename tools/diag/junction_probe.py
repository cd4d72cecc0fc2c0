#!/usr/bin/env python3
"""Stand-alone timing of the fused residual junction + conv1 GEMM (diga_conv2d_junction_f32) against the two separate launches it
replaces (diga_bn_apply + diga_conv2d_nhwc_f32), layer3 / layer4 / layer2 shapes of the C2 step (16 images of 97 x 97)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    dev = "cuda"
    P = _lib.ptr
    m = 16 * 97 * 97
    for name, k, cout in (("layer3 1024->256", 1024, 256), ("layer4 2048->512", 2048, 512), ("layer2 512->128", 512, 128)):
        y3 = torch.randn((m, k), device=dev)
        skip = torch.randn((m, k), device=dev).clamp_min(0)
        ab = torch.cat([torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev) * 0.1])
        w = torch.randn((cout, k), device=dev) * 0.03
        x = torch.empty((m, k), device=dev)
        bits = torch.empty((m, k // 8), dtype=torch.uint8, device=dev)
        out = torch.empty((m, cout), device=dev)
        stats = torch.empty(((m + 63) // 64) * 3 * cout, device=dev)
        st = _lib.stream()
        t_f = timed(lambda: _lib.call("diga_conv2d_junction_f32", P(y3), k, P(skip), k, P(ab), P(x), k, P(bits), P(w), P(out), cout, P(stats),
                                      m, k, cout, st))
        x1, b1, o1 = x.clone(), bits.clone(), out.clone()
        t_a = timed(lambda: _lib.call("diga_bn_apply", P(y3), k, P(x), k, P(skip), k, P(ab), m, k, 1, P(bits), st))
        t_g = timed(lambda: _lib.call("diga_conv2d_nhwc_f32", P(x), P(w), None, P(out), 1, 1, m, k, k, 1, m, cout, cout, 1, 1, 1, 1, 0, 0, 1, 1,
                                      P(stats), 0, st))
        same = bool(torch.equal(x1, x) and torch.equal(b1, bits) and torch.equal(o1, out))
        print(f"{name}: fused {t_f:.3f} ms | apply {t_a:.3f} + gemm {t_g:.3f} = {t_a + t_g:.3f} ms | bit-identical {same}")
        del y3, skip, x, bits, out, x1, b1, o1


if __name__ == "__main__":
    main()
