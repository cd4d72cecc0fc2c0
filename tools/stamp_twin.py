#!/usr/bin/env python3
"""Phase breakdown of conv_fwd_x3t8_kernel from in-kernel s_memtime stamps (a STAMPED build of the library, selected with
DIGA_LIB; the product build has no stamps): per block prologue (launch -> first stage landed), K loop, epilogue, and the
loader waves' time in vmcnt waits -- with the operands warm (re-used across launches) or cold (1 GB sweep before).
Diagnostic only.

    DIGA_LIB=/path/to/libdiga_hip_stamp.so python tools/stamp_twin.py [--only l3.conv3]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402

SHAPES = [("l3.conv3 1x1", 256, 1024, 1, 1, 97), ("l3.conv1 1x1", 1024, 256, 1, 1, 97), ("l3.conv2 3x3 d2", 256, 256, 3, 2, 97),
          ("aspp.d6", 2048, 256, 3, 6, 97)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--images", type=int, default=16)
    a = ap.parse_args()
    dev = "cuda"
    os.environ["DIGA_X3T_VARIANT"] = "1"
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
        tiles = ((m + 255) // 256) * ((cout + 127) // 128)
        stamps = torch.zeros(max(tiles * 8, _lib.lib.diga_conv2d_stats_floats(n, hw, hw, cout)), device=dev)
        y = torch.empty((n, hw, hw, cout), device=dev)
        flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)

        def run():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            _lib.call("diga_conv2d_nhwc_twin", _lib.ptr(twin), _lib.ptr(img), None, _lib.ptr(y), n, hw, hw, cin, hw, hw, cout,
                      cout, k, k, 1, 1, -pad, -pad, dil, dil, _lib.ptr(stamps), 11, _lib.stream())
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e)
        for mode in ("warm", "cold"):
            run()
            if mode == "cold":
                flush.add_(1.0)
                torch.cuda.synchronize()
            ms = run()
            d = stamps[:tiles * 8].view(tiles, 8).cpu()
            pro, loop, epi, ks, lwait, ltot = (float(d[:, i].mean()) for i in range(6))
            print(f"{name:16s} {mode}: {ms:.3f} ms, {tiles} tiles; per tile (cycles of s_memtime @100 MHz x?): prologue {pro:8.0f}  "
                  f"K loop {loop:8.0f} ({ks:.0f} steps, {loop / max(ks, 1):6.0f} per step)  epilogue {epi:8.0f}  | loader vmcnt waits {lwait:8.0f} of {ltot:8.0f}")
        del x, w, twin, img, y, stamps, flush
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
