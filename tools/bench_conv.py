#!/usr/bin/env python3
"""Per-shape timing of the conv kernels (forward, backward-data, backward-weight) on the layer
geometries of ResNet-101 DeepLabV2 at the C2 size (16 images of 768x768 -> 193x193 / 97x97 maps).

    python tools/bench_conv.py [--images 16] [--reps 5]
Prints one line per (shape, pass): ms, TFLOP/s, fraction of the 157.3 TFLOP/s fp32 MFMA peak, and the
share of a training step's conv time that shape accounts for (count x time).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402
from diga_amd.model.conv import DigaConv2d  # noqa: E402

PEAK = 157.3          # fp32 MFMA peak; the split-bf16 mode is priced against 2500 / 3 = 833.3 (see --math)
# name, count per forward, Cin, Cout, k, stride, dil, spatial (H=W)
SHAPES = [
    ("stem7x7", 1, 3, 64, 7, 2, 1, 768),
    ("l1.conv1.first", 1, 64, 64, 1, 1, 1, 193), ("l1.conv1", 2, 256, 64, 1, 1, 1, 193),
    ("l1.conv2", 3, 64, 64, 3, 1, 1, 193), ("l1.conv3", 3, 64, 256, 1, 1, 1, 193), ("l1.down", 1, 64, 256, 1, 1, 1, 193),
    ("l2.conv1.first", 1, 256, 128, 1, 2, 1, 193), ("l2.conv1", 3, 512, 128, 1, 1, 1, 97),
    ("l2.conv2", 4, 128, 128, 3, 1, 1, 97), ("l2.conv3", 4, 128, 512, 1, 1, 1, 97), ("l2.down", 1, 256, 512, 1, 2, 1, 193),
    ("l3.conv1.first", 1, 512, 256, 1, 1, 1, 97), ("l3.conv1", 22, 1024, 256, 1, 1, 1, 97),
    ("l3.conv2", 23, 256, 256, 3, 1, 2, 97), ("l3.conv3", 23, 256, 1024, 1, 1, 1, 97), ("l3.down", 1, 512, 1024, 1, 1, 1, 97),
    ("l4.conv1.first", 1, 1024, 512, 1, 1, 1, 97), ("l4.conv1", 2, 2048, 512, 1, 1, 1, 97),
    ("l4.conv2", 3, 512, 512, 3, 1, 4, 97), ("l4.conv3", 3, 512, 2048, 1, 1, 1, 97), ("l4.down", 1, 1024, 2048, 1, 1, 1, 97),
    ("aspp.1x1", 1, 2048, 256, 1, 1, 1, 97), ("aspp.d6", 1, 2048, 256, 3, 1, 6, 97), ("aspp.d12", 1, 2048, 256, 3, 1, 12, 97),
    ("aspp.d18", 1, 2048, 256, 3, 1, 18, 97), ("aspp.d24", 1, 2048, 256, 3, 1, 24, 97),
    ("aspp.bottleneck", 1, 1280, 256, 3, 1, 1, 97), ("head", 1, 256, 19, 1, 1, 1, 97),
]


def timed(fn, reps):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default=None)
    ap.add_argument("--math", default="f32", choices=["f32", "bf16x3"])
    a = ap.parse_args()
    _lib.set_conv_math(1 if a.math == "bf16x3" else 0)
    global PEAK
    PEAK = 2500.0 / 3.0 if a.math == "bf16x3" else 157.3
    dev = "cuda"
    rows, tot = [], {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for name, count, cin, cout, k, stride, dil, hw in SHAPES:
        if a.only and a.only not in name:
            continue
        pad = dil * (k - 1) // 2
        m = DigaConv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=False).to(dev)
        x = torch.randn((a.images, hw, hw, cin), device=dev).permute(0, 3, 1, 2)
        need_dx = cin > 3
        x.requires_grad_(need_dx)
        y = m(x)
        gy = torch.randn_like(y)
        ho = y.shape[-1]
        flops = 2.0 * a.images * ho * ho * cout * cin * k * k

        def run_fwd():
            with torch.no_grad():
                m(x)

        t_f = timed(run_fwd, a.reps)

        # isolate the two backward kernels through the event profiler inside the library
        _lib.call("diga_prof_reset")
        _lib.call("diga_prof_enable", 1)
        for _ in range(a.reps):
            m.weight.grad = None
            if need_dx:
                x.grad = None
            m(x).backward(gy)
        torch.cuda.synchronize()
        _lib.call("diga_prof_enable", 0)
        nd, td = _lib.prof_query("conv_bwd_data")
        nw, tw = _lib.prof_query("conv_bwd_weight")
        t_d = td / nd if nd else 0.0
        t_w = tw / nw if nw else 0.0
        rows.append((name, count, flops, t_f, t_d, t_w))
        tot["fwd"] += count * t_f
        tot["dgrad"] += count * t_d
        tot["wgrad"] += count * t_w
        del m, x, y, gy
        torch.cuda.empty_cache()
    print(f"{'shape':18s} {'cnt':>3s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>6s} {'frac':>5s} | {'dgrad ms':>8s} {'TF/s':>6s} | "
          f"{'wgrad ms':>8s} {'TF/s':>6s} | share fwd/dgrad/wgrad")
    for name, count, flops, t_f, t_d, t_w in rows:
        tf = lambda t: flops / (t * 1e-3) / 1e12 if t > 0 else 0.0  # noqa: E731
        print(f"{name:18s} {count:3d} {flops / 1e9:8.1f} | {t_f:8.3f} {tf(t_f):6.1f} {tf(t_f) / PEAK:5.2f} | {t_d:8.3f} {tf(t_d):6.1f} | "
              f"{t_w:8.3f} {tf(t_w):6.1f} | {100 * count * t_f / tot['fwd']:.1f}% {100 * count * t_d / max(tot['dgrad'], 1e-9):.1f}% "
              f"{100 * count * t_w / max(tot['wgrad'], 1e-9):.1f}%")
    print(f"sum over one forward: fwd {tot['fwd']:.1f} ms, dgrad {tot['dgrad']:.1f} ms, wgrad {tot['wgrad']:.1f} ms")


if __name__ == "__main__":
    main()
