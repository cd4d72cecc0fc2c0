#!/usr/bin/env python3
"""Where a tile of conv_fwd_dma_kernel spends its time: s_memrealtime stamps (100 MHz) of MFMA wave 0 of every block from a
diagnostic build (-DDIGA_PROBE_STAMP, loaded with DIGA_LIB): entry -> first barrier reached -> stage 0 landed -> K loop done
-> accumulators staged -> drained.  Usage (GPU box):
    DIGA_LIB=$PWD/diga_amd/libdiga_probe_STAMP.so DIGA_CONV_WINOGRAD=0 python tools/diag/f32_tile_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402
from diga_amd.model.conv import DigaConv2d  # noqa: E402

_lib.set_conv_math(0)
fn = _lib.lib.diga_probe_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, n, cin, h, cout, k, d in [("1x1 K=256 ->1024", 2, 256, 256, 1024, 1, 1), ("1x1 K=1024 -> 256", 2, 1024, 256, 256, 1, 1),
                                     ("3x3 K=2304", 2, 256, 256, 256, 3, 2)]:
    m = DigaConv2d(cin, cout, k, padding=d * (k // 2), dilation=d, bias=False).cuda()
    x = torch.randn(n, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            m(x)
    torch.cuda.synchronize()
    nblk = min(8192, (n * h * h // 256) * (cout // 128))
    buf = np.zeros(nblk * 6, dtype=np.uint64)
    rc = fn(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    st = buf.reshape(nblk, 6).astype(np.int64)
    d_ = np.diff(st, axis=1) * 10.0          # ns (100 MHz counter)
    tot = (st[:, 5] - st[:, 0]) * 10.0
    names = ["setup->barrier", "wait stage 0", "K loop", "acc->LDS", "drain"]
    print(f"{name}: {nblk} blocks, mean block life {tot.mean() / 1e3:.2f} us; kernel span {(st[:, 5].max() - st[:, 0].min()) * 10 / 1e3:.1f} us")
    print("   " + ", ".join(f"{nm} {d_[:, i].mean() / 1e3:.2f} us" for i, nm in enumerate(names)))
    # gap between a block's end and the next block's entry on the same CU is not visible here; estimate from span
    rounds = nblk / 256.0
    print(f"   rounds {rounds:.1f}: span/rounds = {(st[:, 5].max() - st[:, 0].min()) * 10 / 1e3 / rounds:.2f} us per round")
