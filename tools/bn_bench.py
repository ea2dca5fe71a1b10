#!/usr/bin/env python3
"""batch_normalize (config 4, 2^24 Jacobian points) HIP-event timing; CHEETAH_HIP_LIB selects a build variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE, Engine
eng = Engine(0)
n = 1 << 16
r = eng.synth_scalars(4, n)
k = eng.synth_scalars(4, n, n)
aff, _ = eng.mul_by_generator_batch(r, AFFINE)
jac, _ = eng.multiply_batch(aff, k, AFFINE, JACOBIAN)          # non-trivial z
want, _ = eng.multiply_batch(aff, k, AFFINE, AFFINE)
big = jac.repeat((1 << 24) // n, 1).contiguous()
best = 1e9
for rep in range(4):
    eng._use_torch_stream(); eng.timer_start()
    xy, inf = eng.batch_normalize(big, JACOBIAN)
    best = min(best, eng.timer_stop_ms())
ok = torch.equal(xy[:n].view(torch.int64), want.view(torch.int64)) and torch.equal(xy[-n:].view(torch.int64), want.view(torch.int64))
print("%s batch_normalize 2^24: %.3f ms  %.2f Gpts/s  %.1f GB/s  ok=%s" % (os.environ.get("CHEETAH_HIP_LIB", "default").split("/")[-1], best, (1 << 24) / best / 1e6, (1 << 24) * 241 / best / 1e6, ok), flush=True)
eng.close()
