#!/usr/bin/env python3
"""batch_normalize per-point time at sizes from cache-resident to HBM-streaming (is memory latency what is left?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import AFFINE, JACOBIAN, Engine
eng = Engine(0)
n0 = 1 << 16
r = eng.synth_scalars(4, n0)
k = eng.synth_scalars(4, n0, n0)
aff, _ = eng.mul_by_generator_batch(r, AFFINE)
jac, _ = eng.multiply_batch(aff, k, AFFINE, JACOBIAN)
for lg in (17, 18, 19, 20, 21, 22, 24):
    big = jac.repeat((1 << lg) // n0, 1).contiguous()
    best = 1e9
    for rep in range(6):
        eng._use_torch_stream(); eng.timer_start()
        xy, inf = eng.batch_normalize(big, JACOBIAN)
        best = min(best, eng.timer_stop_ms())
    print("2^%d: %.4f ms  %.2f Gpts/s  (x%.0f to 2^24: %.3f ms)" % (lg, best, (1 << lg) / best / 1e6, 2 ** (24 - lg), best * 2 ** (24 - lg)), flush=True)
eng.close()
