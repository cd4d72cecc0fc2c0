#!/usr/bin/env python3
"""Host -> device time of one C2 step's inputs (x, x_aug, rec_s2t [8,3,768,768] fp32 + labels [8,768,768] int64 = 208 MB) from
page-locked and from pageable host memory: the PCIe-inclusive note of DESIGN section 5 (the path's boundary takes device tensors)."""
import time

import torch

B, H, W = 8, 768, 768
for pinned in (True, False):
    host = [torch.randn((B, 3, H, W)) for _ in range(3)] + [torch.randint(0, 19, (B, H, W), dtype=torch.int64)]
    if pinned:
        host = [t.pin_memory() for t in host]
    dev = [torch.empty_like(t, device="cuda") for t in host]
    nbytes = sum(t.numel() * t.element_size() for t in host)
    for _ in range(2):
        for d, h in zip(dev, host):
            d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        for d, h in zip(dev, host):
            d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{'pinned' if pinned else 'pageable'}: {nbytes / 1e6:.1f} MB in {dt * 1e3:.2f} ms = {nbytes / dt / 1e9:.1f} GB/s")
