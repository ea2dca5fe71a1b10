#!/usr/bin/env python3
"""One host-pointer mul_by_generator call at 2^22 into re-used arrays (for a kernel trace of the pipeline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cheetah_amd import AFFINE, Engine
eng = Engine(0)
lib, h = eng._lib, eng._h
n = 1 << 22
r = eng.synth_scalars(5, n, 1 << 26).cpu().numpy()
out = np.zeros((n, 12), np.uint64); inf = np.zeros(n, np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.cheetah_mul_by_generator_batch(h, r.ctypes.data, out.ctypes.data, inf.ctypes.data, n, AFFINE)
    print("call %d: rc %d  %.2f ms" % (rep, rc, (time.perf_counter() - t0) * 1e3), flush=True)
eng.close()
