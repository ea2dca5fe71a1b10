#!/usr/bin/env python3
"""Reference point for the HBM roofline: what a plain device copy / read-only reduction achieves on this box."""
import torch
for lg in (27, 30, 32):
    n = 1 << lg  # bytes
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.zero_()
    b = torch.empty_like(a)
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("copy %5d MiB: %.3f ms  read+write %.1f GB/s" % (n >> 20, best, 2 * n / best / 1e6), flush=True)
    v = a.view(torch.int64)
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            s = v.sum()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("sum  %5d MiB: %.3f ms  read %.1f GB/s" % (n >> 20, best, n / best / 1e6), flush=True)
