#!/usr/bin/env python3
"""k_random_candidates of a lab build (tools/lab/build.sh rand -DLAB_RAND) timed, and its output checked: every candidate
on the curve, and equal to the numpy + oracle restatement of the construction on a sample (the first successful
attempt of each element).   usage: rand_lab.py <lg> librand.so"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from cheetah_amd import AFFINE, Engine
from oracle import binding as ora
import lab_common as L
import test_gpu_widening as W

lg = int(sys.argv[1])
lib = L.load(sys.argv[2])
lib.lab_random_candidates.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]
eng = Engine(0)
ora.build()
n = (1 << lg) - 29
seed, start = 91, 5
xy = torch.full((n, 12), -1, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ms = L.timed(lambda: lib.lab_random_candidates(seed, start, xy.data_ptr(), n, st))
on = eng.is_on_curve_batch(xy.view(torch.uint64), AFFINE)
m = 3000
cand, tries = W._random_candidates_model(ora, seed, start, m)
head = xy[:m].cpu().numpy().view(np.uint64)
tail_start = n - 500
cand_t, _ = W._random_candidates_model(ora, seed, start + tail_start, 500)
tail = xy[tail_start:].cpu().numpy().view(np.uint64)
print("k_random_candidates 2^%d: %8.3f ms  %8.2f M/s  on curve: %s  first %d equal to the model: %s  last 500: %s  (attempts: mean %.2f max %d)" % (
    lg, ms, n / ms / 1e3, bool(on.all()), m, bool((head == cand).all()), bool((tail == cand_t).all()), tries.mean(), tries.max()), flush=True)
eng.close()
