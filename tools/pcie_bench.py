#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (chunked upload / kernels / download pipeline,
synchronous for the caller) beside the device-resident rate of the same batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cheetah_amd import AFFINE, JACOBIAN, Engine, synth
eng = Engine(0)
def best(f, reps=3):
    b = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b
def h(t):
    return t.view(torch.int64).cpu().numpy().view(np.uint64) if t.dtype == torch.uint64 else t.cpu().numpy()
def dev_ms(f, reps=2):
    f(); torch.cuda.synchronize(); eng._use_torch_stream(); eng.timer_start()
    for _ in range(reps): f()
    return eng.timer_stop_ms() / reps
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
A, B = eng.synth_fp6(1, n, 0), eng.synth_fp6(1, n, n)
a, b = h(A), h(B)
eng.fp6_mul_batch(a[:1000], b[:1000])
t = best(lambda: eng.fp6_mul_batch(a, b))
print("fp6_mul_batch host 2^%d: %.2f ms  %.1f M/s  (%.1f GB/s over PCIe, both directions summed)" % (lg, t * 1e3, n / t / 1e6, n * 144 / t / 1e9))
K = eng.synth_scalars(5, n)
R = eng.synth_scalars(5, n, 1 << 26)
PTS, _ = eng.mul_by_generator_batch(R, AFFINE)
k, r, pts = h(K), h(R), h(PTS)
td = dev_ms(lambda: eng.mul_by_generator_batch(R, AFFINE)) * 1e-3
t = best(lambda: eng.mul_by_generator_batch(r, AFFINE))
print("mul_by_generator_batch 2^%d: host %.2f ms %.2f M/s | device-resident %.2f M/s | ratio %.3f" % (lg, t * 1e3, n / t / 1e6, n / td / 1e6, td / t))
td = dev_ms(lambda: eng.multiply_batch(PTS, K, AFFINE, AFFINE)) * 1e-3
t = best(lambda: eng.multiply_batch(pts, k, AFFINE, AFFINE), 2)
print("multiply_batch 2^%d: host %.2f ms %.2f M/s | device-resident %.2f M/s | ratio %.3f" % (lg, t * 1e3, n / t / 1e6, n / td / 1e6, td / t))
JAC, _ = eng.multiply_batch(PTS[: n // 16], K[: n // 16], AFFINE, JACOBIAN)
BIG = JAC.repeat(16, 1).contiguous()
big = h(BIG)
td = dev_ms(lambda: eng.batch_normalize(BIG, JACOBIAN)) * 1e-3
t = best(lambda: eng.batch_normalize(big, JACOBIAN))
print("batch_normalize_jacobian 2^%d: host %.2f ms %.1f M/s (%.1f GB/s) | device-resident %.1f M/s" % (lg, t * 1e3, n / t / 1e6, n * 241 / t / 1e9, n / td / 1e6))
eng.close()
