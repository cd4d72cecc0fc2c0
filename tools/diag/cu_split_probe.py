#!/usr/bin/env python3
"""Experiment: does partitioning the CUs between the step's two streams (student / main on one set, teacher + weight gradients on
the other) let bandwidth-bound kernels of one stream run under matrix-core kernels of the other?  (The persistent GEMM fills every
CU's registers and LDS, so two unmasked streams only time-slice.)  Streams with a CU mask come from hipExtStreamCreateWithCUMask.

    python tools/diag/cu_split_probe.py --mask-a ffffffff... --mask-b ...   (hex words, 8 x 32 bits = 256 CUs)"""
import argparse
import ctypes
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def masked_stream(hip, words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


PATTERNS = {
    "all": [0xFFFFFFFF] * 8,
    "lo": [0xFFFFFFFF] * 4 + [0] * 4, "hi": [0] * 4 + [0xFFFFFFFF] * 4,
    "even": [0x55555555] * 8, "odd": [0xAAAAAAAA] * 8,
    "lo3q": [0xFFFFFFFF] * 6 + [0] * 2, "hi1q": [0] * 6 + [0xFFFFFFFF] * 2,
    "x3q": [0x77777777] * 8, "x1q": [0x88888888] * 8,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="all")
    ap.add_argument("--b", default="all")
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    from diga_amd import _lib, synthetic, train_step
    from diga_amd.model import seg_model_noaux as sm
    from diga_amd.model.model_noaux import SegModel
    hip = ctypes.CDLL("libamdhip64.so")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sa, sb = masked_stream(hip, PATTERNS[a.a]), masked_stream(hip, PATTERNS[a.b])
    train_step._STREAMS[(0, "teacher")] = sb
    _lib._side_streams[0] = sb
    _lib.set_conv_math(0)
    torch.manual_seed(0)
    student, teacher = SegModel(arch=sm.RESNET101).to(dev), SegModel(arch=sm.RESNET101).to(dev)
    teacher.train()
    tr = train_step.DigaTrainer(student, teacher, rng=random.Random(1))
    batch = synthetic.warmup_batch(1234, 8, 768, 768, block=32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        for it in range(3):
            tr.warmup_step(it, *batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(3, 3 + a.steps):
            out = tr.warmup_step(it, *batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    print(f"CU split a={a.a} b={a.b}: {dt * 1e3:.1f} ms/step, loss {float(out['total']):.4f}")


if __name__ == "__main__":
    main()
