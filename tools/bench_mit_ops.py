#!/usr/bin/env python3
"""Per-shape timing of the MiT kernels at the MiT-B5 / 16 x 768x768 sizes (operands cold: a 1 GB sweep between launches
when --cold):  python tools/bench_mit_ops.py [--cold] [--what gemm,wgrad,attn,ln,dw]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--what", default="gemm,wgrad,attn,ln,dw")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import torch
    from diga_amd import _lib
    from diga_amd.model.networks.MixTransfomer import _Ops
    dev = torch.device("cuda")
    ops = _Ops(dev)
    P = _lib.ptr
    sweep = torch.empty(1 << 28, dtype=torch.float32, device=dev) if a.cold else None

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        tot = 0.0
        for _ in range(a.reps):
            if sweep is not None:
                sweep.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / a.reps * 1e3          # us

    B = 16
    stages = [(B * 192 * 192, 64, 1, 8), (B * 96 * 96, 128, 2, 4), (B * 48 * 48, 320, 5, 2), (B * 24 * 24, 512, 8, 1)]
    what = a.what.split(",")
    for si, (M, C, heads, sr) in enumerate(stages):
        hid = 4 * C
        print(f"== stage {si + 1}: M={M} C={C}")
        if "gemm" in what:
            shapes = [("q/proj", M, C, C, False), ("fc1", M, hid, C, False), ("fc2+res", M, C, hid, True), ("kv", B * 576, 2 * C, C, False)]
            if sr > 1:
                shapes.append(("sr", B * 576, C, sr * sr * C, True))
            for name, m, n, k, f32 in shapes:
                x = torch.randn((m, k), device=dev).half()
                w = torch.randn((n, k), device=dev).half()
                bias = torch.randn(n, device=dev)
                res = torch.randn((m, n), device=dev) if f32 else None
                out = torch.empty((m, n), device=dev, dtype=torch.float32 if f32 else torch.float16)
                us = timed(lambda: ops.gemm(x, w, bias, n, out_f32=f32, residual=res, out=out))
                byt = m * k * 2 + n * k * 2 + m * n * (8 if f32 else 2)
                print(f"  gemm {name:8s} [{m}x{n}x{k}] {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s  {byt / us / 1e3:7.1f} GB/s")
        if "wgrad" in what:
            for name, m, n, k in (("q/proj", M, C, C), ("fc1", M, hid, C), ("fc2", M, C, hid)):
                dy = torch.randn((m, n), device=dev).half()
                x = torch.randn((m, k), device=dev).half()
                us = timed(lambda: ops.wgrad(dy, x, 1.0, bias=True))
                print(f"  wgrad {name:7s} [{m}: {n}x{k}] {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s  {(m * (n + k) * 2) / us / 1e3:7.1f} GB/s")
        if "attn" in what:
            N, nk = M // B, 576
            q = torch.randn((M, C), device=dev).half()
            kv = torch.randn((B * nk, 2 * C), device=dev).half()
            o = torch.empty_like(q)
            lse = torch.empty((B, heads, N), device=dev)
            us = timed(lambda: _lib.call("diga_mit_attention_fwd", P(q), C, P(kv), 2 * C, P(o), C, P(lse), B, heads, N, nk, 0.125, _lib.stream()))
            fl = 4.0 * B * heads * N * nk * 64
            print(f"  attn fwd {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s")
            do = torch.randn_like(q)
            dq, dkv = torch.empty_like(q), torch.empty_like(kv)
            ws = torch.empty(_lib.lib.diga_mit_attention_bwd_workspace_bytes(B, heads, N, nk), dtype=torch.uint8, device=dev)
            us = timed(lambda: _lib.call("diga_mit_attention_bwd", P(q), C, P(kv), 2 * C, P(o), P(do), C, P(lse), P(dq), P(dkv), P(ws), ws.numel(),
                                         B, heads, N, nk, 0.125, _lib.stream()))
            print(f"  attn bwd {us:8.1f} us {14.0 / 4 * fl / us / 1e6:7.1f} TFLOP/s")
        if "ln" in what:
            x = torch.randn((M, C), device=dev)
            g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            us = timed(lambda: ops.ln_fwd(x, g, b, 1e-6))
            print(f"  ln fwd {us:8.1f} us {M * C * 6 / us / 1e3:7.1f} GB/s")
            _, _, mean, rstd = ops.ln_fwd(x, g, b, 1e-6)
            dy = torch.randn((M, C), device=dev).half()
            us = timed(lambda: ops.ln_bwd(dy, x, g, mean, rstd, x, True, True, 1.0))
            print(f"  ln bwd {us:8.1f} us {M * C * 16 / us / 1e3:7.1f} GB/s")
        if "dw" in what:
            hw = int((M // B) ** 0.5)
            x = torch.randn((B, hw, hw, hid), device=dev).half()
            wt9 = torch.randn((9, hid), device=dev)
            bias = torch.randn(hid, device=dev)
            u, h = torch.empty_like(x), torch.empty_like(x)
            us = timed(lambda: _lib.call("diga_mit_dwconv_gelu_fwd", P(x), P(wt9), P(bias), P(u), P(h), B, hw, hw, hid, _lib.stream()))
            print(f"  dwconv fwd {us:8.1f} us {M * hid * 6 / us / 1e3:7.1f} GB/s")
            du, dx = torch.empty_like(x), torch.empty_like(x)
            dw, db = torch.empty((hid, 9), device=dev), torch.empty(hid, device=dev)
            ws = torch.empty(_lib.lib.diga_mit_dwconv_bwd_workspace_bytes(B, hw, hid), dtype=torch.uint8, device=dev)
            us = timed(lambda: _lib.call("diga_mit_dwconv_gelu_bwd", P(h), P(u), P(x), P(wt9), P(du), P(dx), P(dw), P(db), 1.0, 0, P(ws),
                                         ws.numel(), B, hw, hw, hid, _lib.stream()))
            print(f"  dwconv bwd {us:8.1f} us {M * hid * 12 / us / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
