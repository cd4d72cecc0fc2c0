"""Timing of diga_mit_gemm_nt on the MiT-B5 shapes (and, with a library built from tools/experiments/mit_gemm_persistent_fp16.hip,
the A/B of its variants through diga_mit_debug_nt_wide: 1 = 256x256 tile, 2 = persistent, +4 no loads, +8 no MFMA, +16 no epilogue): time per call (back to back and with a 512 MB write in between =
cold caches), TFLOP/s, max error against torch.  Usage: python tools/diag/gemm_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402
from diga_amd.model.networks.MixTransfomer import _Ops  # noqa: E402

ops = _Ops(torch.device('cuda'))
flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')


def t(m, n, k, f32=False, reps=10, cold=False):
    x = torch.randn((m, k), device='cuda').half()
    w = (torch.randn((n, k), device='cuda') / k ** 0.5).half()
    out = torch.empty((m, n), device='cuda', dtype=torch.float32 if f32 else torch.float16)
    ops.gemm(x, w, None, n, out_f32=f32, out=out)
    ref = (x[:2048].float() @ w.float().t())
    err = float((out[:2048].float() - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if cold:
            flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(x, w, None, n, out_f32=f32, out=out)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    us = tot / reps * 1e3
    return us, 2.0 * m * n * k / us / 1e6, err


shapes = [(36864, 1280, 320), (147456, 512, 128), (36864, 1280, 1280), (8192, 8192, 8192)]
_unused = [(36864, 1280, 320), (36864, 320, 1280), (36864, 320, 320), (147456, 512, 128), (147456, 128, 512), (9216, 2048, 512),
          (9216, 512, 2048), (589824, 256, 64), (589824, 64, 256), (147456, 128, 128), (36864, 1280, 1280), (8192, 8192, 8192)]
for cold in (False,):
    print("== cold caches" if cold else "== back to back")
    for m, n, k in shapes:
        row = f"  [{m}x{n}x{k}]"
        for wide in ((0, 2, 2 + 4, 2 + 16, 2 + 4 + 16) if hasattr(_lib.lib, "diga_mit_debug_nt_wide") else (0,)):
            if wide or hasattr(_lib.lib, "diga_mit_debug_nt_wide"):
                _lib.lib.diga_mit_debug_nt_wide(wide)
            us, tf, err = t(m, n, k, cold=cold)
            row += f"  m{wide}: {us:7.1f}us {tf:5.0f}TF"
        print(row)
