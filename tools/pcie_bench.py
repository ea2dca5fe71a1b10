#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (copy in, kernels, copy out, synchronous)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cheetah_amd import AFFINE, JACOBIAN, Engine, synth
eng = Engine(0)
def best(f, reps=3):
    b = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b
n = 1 << 20
a, b = synth.fp6_elements(1, n, 0), synth.fp6_elements(1, n, n)
t = best(lambda: eng.fp6_mul_batch(a, b))
print("fp6_mul_batch host 2^20: %.2f ms  %.1f M/s  (%.1f GB/s over PCIe)" % (t * 1e3, n / t / 1e6, n * 144 / t / 1e9))
m = 1 << 18
k = synth.scalars(5, m)
r = synth.scalars(5, m, 1 << 26)
pts, _ = eng.mul_by_generator_batch(r, AFFINE)
t = best(lambda: eng.mul_by_generator_batch(r, AFFINE))
print("mul_by_generator_batch host 2^18: %.2f ms  %.2f M/s" % (t * 1e3, m / t / 1e6))
t = best(lambda: eng.multiply_batch(pts, k, AFFINE, AFFINE))
print("multiply_batch host 2^18: %.2f ms  %.2f M/s" % (t * 1e3, m / t / 1e6))
jac, _ = eng.multiply_batch(pts, k, AFFINE, JACOBIAN)
big = np.tile(jac, (16, 1))
t = best(lambda: eng.batch_normalize(big, JACOBIAN))
print("batch_normalize_jacobian host 2^22: %.2f ms  %.1f M/s" % (t * 1e3, len(big) / t / 1e6))
eng.close()
