#!/usr/bin/env python3
"""A/B timing of the staging-free split-bf16 forward kernel variants (DIGA_X3T_VARIANT) on C2 layer shapes, interleaved
rounds in one process (cdna_hip_programming.md rule 24), plus a bit-identity check between the variants.

    python tools/bench_twin.py [--variants 0 1] [--rounds 5] [--only aspp]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402

# name, Cin, Cout, k, dil, H=W (16 images)
SHAPES = [
    ("l3.conv2 3x3 d2", 256, 256, 3, 2, 97), ("l4.conv2 3x3 d4", 512, 512, 3, 4, 97), ("aspp.d6", 2048, 256, 3, 6, 97),
    ("aspp.d24", 2048, 256, 3, 24, 97), ("aspp.bottleneck", 1280, 256, 3, 1, 97), ("l3.conv1 1x1", 1024, 256, 1, 1, 97),
    ("l3.conv3 1x1", 256, 1024, 1, 1, 97), ("l4.conv1 1x1", 2048, 512, 1, 1, 97), ("l4.conv3 1x1", 512, 2048, 1, 1, 97),
    ("l2.conv2 3x3", 128, 128, 3, 1, 97), ("l1.conv2 3x3", 64, 64, 3, 1, 193),
    ("l2.conv1 1x1", 512, 128, 1, 1, 97), ("l2.conv3 1x1", 128, 512, 1, 1, 97), ("l1.conv3 1x1", 64, 256, 1, 1, 193),
    ("l3.conv1.dgrad 1x1", 256, 1024, 1, 1, 97), ("l4.conv1.dgrad 1x1", 512, 2048, 1, 1, 97),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, nargs="+", default=[0, 1],
                    help="DIGA_X3T_VARIANT values (0: 8-wave kernel, 1: 12-wave kernel)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--only", default=None)
    ap.add_argument("--stats", action="store_true", help="with the BatchNorm-partials epilogue")
    ap.add_argument("--cold", action="store_true",
                    help="sweep a 1 GB buffer before every timed launch, so that the conv reads its input from HBM as it does "
                         "inside a training step (a 154 MB twin otherwise stays in the 256 MB Infinity Cache across launches)")
    a = ap.parse_args()
    dev = "cuda"
    print(f"{'shape':20s} " + " ".join(f"{'v%d ms' % v:>9s} {'TF/s':>6s}" for v in a.variants) + "   identical")
    for name, cin, cout, k, dil, hw in SHAPES:
        if a.only and a.only not in name:
            continue
        n = a.images
        pad = dil * (k // 2)
        x = torch.randn((n, hw, hw, cin), device=dev)
        w = torch.randn((cout, k, k, cin), device=dev) * (2.0 / (cin * k * k)) ** 0.5
        m = n * hw * hw
        twin = torch.empty(m * cin * 4, dtype=torch.uint8, device=dev)
        _lib.call("diga_make_twin", _lib.ptr(x), cin, _lib.ptr(twin), m, cin, _lib.stream())
        img = torch.empty(_lib.lib.diga_split_bf16_image_bytes(cout, k * k, cin), dtype=torch.uint8, device=dev)
        _lib.call("diga_split_bf16_image", _lib.ptr(w), _lib.ptr(img), cout, k * k, cin, _lib.stream())
        stats = torch.empty(_lib.lib.diga_conv2d_stats_floats(n, hw, hw, cout), dtype=torch.float32, device=dev) if a.stats else None
        outs = {v: torch.empty((n, hw, hw, cout), device=dev) for v in a.variants}

        def run(v):
            os.environ["DIGA_X3T_VARIANT"] = str(v)
            _lib.call("diga_conv2d_nhwc_twin", _lib.ptr(twin), _lib.ptr(img), None, _lib.ptr(outs[v]), n, hw, hw, cin, hw, hw, cout,
                      cout, k, k, 1, 1, -pad, -pad, dil, dil, _lib.ptr(stats), 11, _lib.stream())

        for v in a.variants:
            run(v)
        torch.cuda.synchronize()
        ref = outs[a.variants[0]]
        same = " ".join("=" if torch.equal(ref, outs[v]) else f"{float((ref - outs[v]).abs().max() / ref.abs().max()):.1e}" for v in a.variants[1:])
        best = {v: 1e9 for v in a.variants}
        flush = torch.empty(256 << 20, dtype=torch.float32, device=dev) if a.cold else None
        for _ in range(a.rounds):
            for v in a.variants:
                if a.cold:
                    tot = 0.0
                    for _ in range(a.reps):
                        flush.add_(1.0)                    # 1 GB read + 1 GB written: nothing of the operands survives
                        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        s.record()
                        run(v)
                        e.record()
                        torch.cuda.synchronize()
                        tot += s.elapsed_time(e)
                    best[v] = min(best[v], tot / a.reps)
                    continue
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.reps):
                    run(v)
                e.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], s.elapsed_time(e) / a.reps)
        flops = 2.0 * m * cout * cin * k * k
        print(f"{name:20s} " + " ".join(f"{best[v]:9.3f} {flops / best[v] / 1e9:6.0f}" for v in a.variants) + f"   {same}")
        del x, w, twin, img, outs, flush
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
