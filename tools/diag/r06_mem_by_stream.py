#!/usr/bin/env python3
"""Where `reserved` lives: torch.cuda.memory_snapshot() after a few C2 (or c4) steps, segments summed per allocation STREAM.
    python tools/diag/r06_mem_by_stream.py [c2|c4] [batch]"""
import collections
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from diga_amd import _lib, synthetic  # noqa: E402
from diga_amd.model.model_noaux import SegModel  # noqa: E402
from diga_amd.train_step import DigaTrainer, _STREAMS  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    student, teacher = SegModel().to(dev), SegModel().to(dev)
    teacher.train()
    tr = DigaTrainer(student, teacher, rng=random.Random(1))
    if cfg == "c4":
        from diga_amd.calc_centroids import Class_Features
        batch = synthetic.selftrain_batch(1234, B, 512, 1024, block=32, device=dev)
        cf = Class_Features(numbers=19)
        cf.objective_vectors = torch.randn((19, 256)).to(dev)
        step = lambda i: tr.selftrain_step(i, *batch, cf)  # noqa: E731
    else:
        batch = synthetic.warmup_batch(1234, B, 768, 768, block=32, device=dev)
        step = lambda i: tr.warmup_step(i, *batch)  # noqa: E731
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    names = {torch.cuda.default_stream(dev).cuda_stream: "main (default)"}
    for (idx, role), st in _STREAMS.items():
        names[st.cuda_stream] = role
    for idx, st in _lib._side_streams.items():
        names[st.cuda_stream] = "wgrad side"
    per = collections.defaultdict(lambda: [0, 0, 0, collections.Counter()])
    for seg in torch.cuda.memory_snapshot():
        p = per[seg["stream"]]
        p[0] += seg["total_size"]
        p[1] += seg["allocated_size"]
        p[2] += 1
        p[3][round(seg["total_size"] / 2 ** 30, 2)] += 1
    print(f"{cfg} B={B}: reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB (peak {torch.cuda.max_memory_reserved() / 1e9:.1f}), allocated now "
          f"{torch.cuda.memory_allocated() / 1e9:.1f} GB (peak {torch.cuda.max_memory_allocated() / 1e9:.1f})")
    for st, (tot, alloc, n, sizes) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        big = ", ".join(f"{k} GiB x{v}" for k, v in sorted(sizes.items(), reverse=True)[:6])
        print(f"  stream {names.get(st, hex(st)):16s} segments {n:4d}  reserved {tot / 1e9:7.2f} GB  live now {alloc / 1e9:7.2f} GB   largest segments: {big}")
    ws = collections.defaultdict(int)
    for (d, tag, st), buf in _lib._workspaces.items():
        ws[(names.get(st, hex(st)), tag)] += buf.numel()
    print("  grow-only workspaces (GB): " + ", ".join(f"{k[0]}/{k[1]} {v / 1e9:.2f}" for k, v in sorted(ws.items(), key=lambda kv: -kv[1])[:12]))


if __name__ == "__main__":
    main()
