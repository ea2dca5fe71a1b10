#!/usr/bin/env python3
"""Host-pointer entry points with the caller's output arrays already faulted in (a caller that re-uses its buffers),
beside the same call into freshly allocated arrays (what tools/pcie_bench.py measures through Engine)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cheetah_amd import AFFINE, JACOBIAN, Engine
eng = Engine(0)
lib, h = eng._lib, eng._h
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
def hst(t):
    return t.view(torch.int64).cpu().numpy().view(np.uint64) if t.dtype == torch.uint64 else t.cpu().numpy()
def best(f, reps=3):
    b = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); rc = f(); b = min(b, time.perf_counter() - t0)
        assert rc == 0
    return b
r = hst(eng.synth_scalars(5, n, 1 << 26))
out = np.zeros((n, 12), np.uint64); inf = np.zeros(n, np.uint8)
t = best(lambda: lib.cheetah_mul_by_generator_batch(h, r.ctypes.data, out.ctypes.data, inf.ctypes.data, n, AFFINE))
print("mul_by_generator_batch 2^%d, outputs pre-faulted: %.2f ms  %.1f M/s (%.1f GB/s)" % (lg, t * 1e3, n / t / 1e6, n * 129 / t / 1e9))
a, b = hst(eng.synth_fp6(1, n, 0)), hst(eng.synth_fp6(1, n, n))
o = np.zeros((n, 6), np.uint64)
t = best(lambda: lib.cheetah_fp6_mul_batch(h, a.ctypes.data, b.ctypes.data, o.ctypes.data, n))
print("fp6_mul_batch 2^%d, outputs pre-faulted: %.2f ms  %.1f M/s (%.1f GB/s)" % (lg, t * 1e3, n / t / 1e6, n * 144 / t / 1e9))
pts = out.copy()
k = hst(eng.synth_scalars(5, n))
t = best(lambda: lib.cheetah_multiply_batch(h, pts.ctypes.data, AFFINE, k.ctypes.data, out.ctypes.data, inf.ctypes.data, n, AFFINE), 2)
print("multiply_batch 2^%d, outputs pre-faulted: %.2f ms  %.2f M/s" % (lg, t * 1e3, n / t / 1e6))
# raw copies of the same sizes for scale
x = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
hp = torch.empty(n * 96, dtype=torch.uint8).pin_memory()
hn = torch.zeros(n * 96, dtype=torch.uint8)
for name, dst, src in (("D2H pinned", hp, x), ("D2H pageable", hn, x), ("H2D pinned", x, hp), ("H2D pageable", x, hn)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(src); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(src); torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    print("%-13s %d MB: %.1f GB/s" % (name, n * 96 >> 20, n * 96 / dt / 1e9))
t0 = time.perf_counter(); hn2 = hn.clone(); dt = time.perf_counter() - t0
print("host memcpy (1 thread, fresh destination) %.1f GB/s" % (n * 96 / dt / 1e9))
t0 = time.perf_counter(); hn2.copy_(hn); dt = time.perf_counter() - t0
print("host memcpy (torch, faulted destination) %.1f GB/s" % (n * 96 / dt / 1e9))
eng.close()
