#!/usr/bin/env python3
"""HIP-event timings of round 3's widened operations: seeded random points, Fp sqrt / exp, Fp6 exp, Fp6 sqrt.
usage: widening_bench.py [lg, default 20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE, Engine
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = Engine(0)
n = 1 << lg
P = 0xFFFFFFFF00000001
def timed(name, f, units, reps=3):
    f()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        eng._use_torch_stream(); eng.timer_start()
        f()
        best = min(best, eng.timer_stop_ms())
    print("%-52s 2^%d: %9.3f ms  %9.3f M/s" % (name, lg, best, units / best / 1e3), flush=True)
for form, nm in ((AFFINE, "affine"), (PROJECTIVE, "projective"), (JACOBIAN, "jacobian")):
    timed("random_points -> %s" % nm, lambda: eng.random_points(9, n, 0, form), n)
pts, _ = eng.random_points(9, n, 0, AFFINE)
timed("clear_cofactor (multiply_vartime by h, 130 bits)", lambda: eng.clear_cofactor_batch(pts, AFFINE, AFFINE), n)
timed("is_torsion_free (multiply_vartime by q, 255 bits)", lambda: eng.is_torsion_free_batch(pts, AFFINE), n)
a = eng.synth_fp6(3, n)
fp = a.view(torch.int64).reshape(-1)[: n].contiguous().view(torch.uint64)
timed("Fp::sqrt", lambda: eng.fp_op_batch("sqrt", fp), n)
timed("Fp::exp (64-bit exponent)", lambda: eng.fp_op_batch("exp", fp, scalar=P - 2), n)
timed("Fp::invert", lambda: eng.fp_op_batch("invert", fp), n)
timed("Fp6::exp (p^6 - 2: 384 bits)", lambda: eng.fp6_exp_batch(a, P ** 6 - 2), n)
timed("Fp6::invert", lambda: eng.fp6_invert_batch(a), n)
timed("Fp6::sqrt", lambda: eng.fp6_sqrt_batch(a), n)
eng.close()
