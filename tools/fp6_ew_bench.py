#!/usr/bin/env python3
"""HIP-event timings of the elementwise Fp6 kernels (BASELINE config 1 and larger), one GPU:
20 back-to-back launches per measurement, best of 3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import Engine
eng = Engine(0)
for lg in (20, 22, 24):
    n = 1 << lg
    a, b = eng.synth_fp6(1, n, 0), eng.synth_fp6(1, n, n)
    for name, f, byt in (("mul", lambda: eng.fp6_mul_batch(a, b), 144), ("square", lambda: eng.fp6_square_batch(a), 96)):
        best = 1e30
        f()
        for _ in range(3):
            eng._use_torch_stream(); eng.timer_start()
            for _ in range(20):
                f()
            best = min(best, eng.timer_stop_ms() / 20)
        print("fp6_%-6s 2^%d: %8.4f ms  %7.2f G/s  %7.1f GB/s" % (name, lg, best, n / best / 1e6, n * byt / best / 1e6), flush=True)
eng.close()
