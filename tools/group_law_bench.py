#!/usr/bin/env python3
"""Complete additions / doublings over a batch (cheetah_add_batch_dev / cheetah_double_batch_dev), HIP-event timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE, Engine
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
eng = Engine(0)
n = 1 << 16
r = eng.synth_scalars(4, n)
k = eng.synth_scalars(4, n, n)
aff, _ = eng.mul_by_generator_batch(r, AFFINE)
rep = (1 << lg) // n
for form, name in ((PROJECTIVE, "projective"), (JACOBIAN, "jacobian")):
    p, _ = eng.multiply_batch(aff, k, AFFINE, form)          # non-trivial z
    q, _ = eng.multiply_batch(aff, r, AFFINE, form)
    P, Q, A = p.repeat(rep, 1).contiguous(), q.repeat(rep, 1).contiguous(), aff.repeat(rep, 1).contiguous()
    out = torch.empty_like(P)
    for what, f, byts in (("add", lambda: eng.add_batch(P, Q, form, out=out), 432), ("add_mixed", lambda: eng.add_batch(P, A, form, AFFINE, out=out), 384),
                          ("double", lambda: eng.double_batch(P, form, out=out), 288)):
        best = 1e9
        for _ in range(4):
            eng._use_torch_stream(); eng.timer_start()
            f()
            best = min(best, eng.timer_stop_ms())
        print("%-10s %-9s 2^%d: %.3f ms  %.2f G/s  %.0f GB/s" % (name, what, lg, best, (1 << lg) / best / 1e6, (1 << lg) * byts / best / 1e6), flush=True)
eng.close()
