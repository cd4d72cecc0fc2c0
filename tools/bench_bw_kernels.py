#!/usr/bin/env python3
"""The HBM-bound kernels north_star names, each timed on its own at full size (bench.py::bandwidth_kernels + the in-step ones:
ClassMix histogram / paste, the fused upsample + CE + distillation block at C2 size):  python tools/bench_bw_kernels.py [out.json]
Prints kernel, launch time, algorithmic bytes, fraction of 8 TB/s."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from diga_amd import _lib  # noqa: E402
from diga_amd.util import loss as L  # noqa: E402
from diga_amd.util import utils as U  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    out = bench.bandwidth_kernels(dev)

    def timed(tags, fn, note):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        _lib.call("diga_prof_reset")
        _lib.call("diga_prof_enable", 1)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        _lib.call("diga_prof_enable", 0)
        for tag in tags:
            n, ms = _lib.prof_query(tag)
            if n:
                nbytes = _lib.prof_work(tag) / n
                out[tag] = {"achieved": nbytes / (ms / n * 1e-3) / 1e9, "frac": nbytes / (ms / n * 1e-3) / 1e9 / 8000.0,
                            "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms / n, "launches": n, "shape": note}
        _lib.call("diga_prof_reset")

    B, H, W = 8, 768, 768
    g = torch.Generator().manual_seed(3)
    lab = torch.randint(0, 19, (B, H // 32, W // 32), generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2).to(dev)
    a, b = torch.randn((B, 3, H, W), device=dev), torch.randn((B, 3, H, W), device=dev)
    import random
    rng = random.Random(1)
    timed(["classmix_hist", "classmix_paste"], lambda: U.classmix(a, b, lab, rng), f"ClassMix on [{B},3,{H},{W}] fp32 + int64 labels")
    s_lr = torch.randn((2 * B, 19, 97, 97), device=dev).requires_grad_()
    t_lr = torch.randn((2 * B, 19, 97, 97), device=dev)
    timed(["upsample_loss"], lambda: L.upsample_ce_distill(s_lr, t_lr, lab, 1.0, 0.5, 0.5), f"upsample + CE + distill fwd+bwd, logits [{2 * B},19,97,97] -> {H}x{W}")
    for k, v in out.items():
        print(f"{k:18s} {1e3 * v['avg_launch_ms']:9.1f} us  {v['algorithmic_bytes_per_launch'] / 1e6:9.1f} MB  frac of 8 TB/s {v['frac']:.3f}")
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "bw_kernels.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
