"""Times the SegFormer decode head alone at the c5 geometry (16 crops of 768x768 -> stage maps 192/96/48/24): forward without
grad (the teacher's share), forward + backward (the student's).  Usage: python tools/diag/segformer_head_probe.py [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd import _lib  # noqa: E402
from diga_amd.model.networks.segformer_head import SegFormerHead  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
chans = [64, 128, 320, 512]
h = SegFormerHead(in_channels=chans, channels=128, feature_strides=[4, 8, 16, 32], num_classes=19, in_index=[0, 1, 2, 3]).to(dev).train()
feats = [torch.randn((B, c, 768 // s, 768 // s), device=dev).contiguous(memory_format=torch.channels_last) for c, s in zip(chans, (4, 8, 16, 32))]


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def fwd():
    with torch.no_grad():
        h(feats)


def fwd_bwd():
    fs = [f.detach().requires_grad_() for f in feats]
    out, _ = h(fs)
    out.backward(torch.ones_like(out))


print("head forward (no grad) ms:", round(timed(fwd), 3))
print("head forward + backward ms:", round(timed(fwd_bwd), 3))
_lib.lib.diga_prof_enable(1)
_lib.prof_reset() if hasattr(_lib, "prof_reset") else None
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
    fwd_bwd()
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=90, max_src_column_width=0))
