#!/usr/bin/env python3
"""MiT-B5 encoder forward + backward on synthetic 768x768 crops (the profiling harness behind profiles/r03_mit_*):
    python tools/bench_mit.py [--batch 16] [--steps 3] [--arch mit_b5] [--fwd-only]
prints per-family HIP-event times (diga_prof_*) and ms per pass."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, nargs=2, default=[768, 768])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--arch", default="mit_b5")
    ap.add_argument("--fwd-only", action="store_true")
    a = ap.parse_args()
    import torch
    from diga_amd import _lib
    from diga_amd.model.networks import MixTransfomer as M
    torch.manual_seed(0)
    m = getattr(M, a.arch)().cuda().eval()
    x = torch.rand((a.batch, 3, *a.size), device="cuda") * 2 - 1

    def step():
        if a.fwd_only:
            with torch.no_grad():
                return m(x)
        for p in m.parameters():
            p.grad = None
        outs = m(x)
        (outs[3].float().sum() * 1e-3).backward()

    step()
    torch.cuda.synchronize()
    _lib.call("diga_prof_reset")
    _lib.call("diga_prof_enable", 1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    _lib.call("diga_prof_enable", 0)
    fam = {}
    for tag in _lib.PROF_TAGS:
        n, ms = _lib.prof_query(tag)
        if n:
            w = _lib.prof_work(tag)
            fam[tag] = {"launches_per_pass": n / a.steps, "ms_per_pass": ms / a.steps, "work_per_pass": w / a.steps,
                        "rate": (w / a.steps) / (ms / a.steps * 1e-3)}
    print(json.dumps({"arch": a.arch, "batch": a.batch, "size": a.size, "fwd_only": a.fwd_only, "ms_per_pass": dt * 1e3,
                      "kernel_ms_per_pass": sum(v["ms_per_pass"] for v in fam.values()), "families": fam}))


if __name__ == "__main__":
    main()
