#!/usr/bin/env python3
"""Diagnostic: phase breakdown of the wide split-bf16 conv kernel from in-kernel s_memtime stamps.

    DIGA_LIB=$PWD/diga_amd/libdiga_probe_STAMP.so python tools/stamp_conv.py [--cin 1024 --cout 256 --k 1 --dil 1 --hw 97 --images 16]
Prints average cycles per K-step (wave 0 of every block) spent in: loads issue + MFMA, barrier 1, waiting for the
global loads, split + ds_write, barrier 2.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=1024)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--dil", type=int, default=1)
    ap.add_argument("--hw", type=int, default=97)
    ap.add_argument("--images", type=int, default=16)
    a = ap.parse_args()
    assert "probe_STAMP" in os.environ.get("DIGA_LIB", ""), "load the stamped build: DIGA_LIB=.../libdiga_probe_STAMP.so (tools/diag/build_probe.sh)"
    dev = "cuda"
    n, h, w = a.images, a.hw, a.hw
    x = torch.randn((n, h, w, a.cin), device=dev)
    wt = torch.randn((a.cout, a.k, a.k, a.cin), device=dev) * 0.05
    hi = torch.empty(wt.numel(), dtype=torch.int16, device=dev)
    lo = torch.empty_like(hi)
    _lib.call("diga_split_bf16", _lib.ptr(wt), _lib.ptr(hi), _lib.ptr(lo), wt.numel(), _lib.stream())
    out = torch.empty((n, h, w, a.cout), device=dev)
    m = n * h * w
    blocks = ((m + 255) // 256) * ((a.cout + 127) // 128)
    dbg = torch.zeros(max(blocks * 8, _lib.lib.diga_conv2d_stats_floats(n, h, w, a.cout)), device=dev)
    pad = a.dil * (a.k // 2)
    for _ in range(3):
        _lib.call("diga_conv2d_nhwc_bf16x3", _lib.ptr(x), _lib.ptr(hi), _lib.ptr(lo), None, _lib.ptr(out), n, h, w, a.cin,
                  a.cin, h, w, a.cout, a.cout, a.k, a.k, 1, 1, -pad, -pad, a.dil, a.dil, _lib.ptr(dbg), 0, _lib.stream())
    torch.cuda.synchronize()
    d = dbg[:blocks * 8].view(blocks, 8).cpu().double()
    steps = d[:, 5].clamp(min=1)
    names = ["loads+mfma issue", "barrier 1", "wait global loads", "split + ds_write", "barrier 2"]
    per = (d[:, :5] / steps[:, None])
    print(f"blocks {blocks}, K-steps {int(steps[0])}")
    for i, nm in enumerate(names):
        print(f"  {nm:20s} mean {per[:, i].mean():8.0f}  p10 {per[:, i].quantile(0.1):8.0f}  p90 {per[:, i].quantile(0.9):8.0f} cycles/step")
    print(f"  total {per.sum(1).mean():8.0f} cycles/step (MFMA pipe time of one wave: 1536)")


if __name__ == "__main__":
    main()
