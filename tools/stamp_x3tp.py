#!/usr/bin/env python3
"""Diagnostic: where an MFMA wave of the persistent pointwise kernel spends its cycles (in-kernel s_memtime stamps).

    DIGA_X3TP_STAMP=1 python tools/stamp_x3tp.py [--cin 256 --cout 1024 --hw 97 --images 16]
Per tile (wave 0 of every block): K-loop time up to the end-of-step barrier, time inside those barriers (= waiting for the
loader waves / the other MFMA waves), epilogue."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=97)
    ap.add_argument("--images", type=int, default=16)
    a = ap.parse_args()
    assert os.environ.get("DIGA_X3TP_STAMP"), "set DIGA_X3TP_STAMP=1"
    dev = "cuda"
    n, hw, cin, cout = a.images, a.hw, a.cin, a.cout
    m = n * hw * hw
    x = torch.randn((m, cin), device=dev)
    w = torch.randn((cout, 1, 1, cin), device=dev) * 0.05
    twin = torch.empty(m * cin * 4, dtype=torch.uint8, device=dev)
    _lib.call("diga_make_twin", _lib.ptr(x), cin, _lib.ptr(twin), m, cin, _lib.stream())
    img = torch.empty(_lib.lib.diga_split_bf16_image_bytes(cout, 1, cin), dtype=torch.uint8, device=dev)
    _lib.call("diga_split_bf16_image", _lib.ptr(w), _lib.ptr(img), cout, 1, cin, _lib.stream())
    out = torch.empty((m, cout), device=dev)
    dbg = torch.zeros(max(256 * 8, _lib.lib.diga_conv2d_stats_floats(n, hw, hw, cout)), device=dev)
    for _ in range(3):
        _lib.call("diga_conv2d_nhwc_twin", _lib.ptr(twin), _lib.ptr(img), None, _lib.ptr(out), n, hw, hw, cin, hw, hw, cout, cout, 1, 1,
                  1, 1, 0, 0, 1, 1, _lib.ptr(dbg), 11, _lib.stream())
    torch.cuda.synchronize()
    d = dbg[:256 * 8].view(256, 8).cpu().double()
    d = d[d[:, 4] > 0]
    tiles, steps = d[:, 4], d[:, 5]
    print(f"blocks {len(d)}, tiles per block {tiles.mean():.1f}, K-steps per tile {int(steps[0])}")
    for i, nm in enumerate(["K-loop (reads + MFMA issue)", "in the per-step barrier", "epilogue"]):
        per = d[:, i] / tiles
        print(f"  {nm:30s} {per.mean():9.0f} cycles per tile  ({(per / steps).mean():7.0f} per K-step)")
    print(f"  total {(d[:, 3] / tiles).mean():9.0f} cycles per tile; MFMA pipe time of one wave per tile: {int(steps[0]) * 1536}")


if __name__ == "__main__":
    main()
