#!/usr/bin/env python3
"""HIP-event timings of the ladder kernels at BASELINE configs 2 / 3 / 5 sizes (one GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE, Engine
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
eng = Engine(0)
if len(sys.argv) > 2:
    eng.set_launch_rounds(int(sys.argv[2]))   # rounds of the resident grid per launch (library default 2)
n = 1 << lg
r = eng.synth_scalars(5, n, 1 << 26)
k = eng.synth_scalars(5, n, 0)
def timed(name, f, reps=2):
    best = 1e30
    for _ in range(reps):
        eng._use_torch_stream(); eng.timer_start()
        out = f()
        best = min(best, eng.timer_stop_ms())
    print("%-44s 2^%d: %9.3f ms  %8.3f M/s" % (name, lg, best, n / best / 1e3), flush=True)
    return out
pts, _ = timed("mul_by_generator BASEPOINT_TABLE affine", lambda: eng.mul_by_generator_batch(r, AFFINE))
timed("mul_by_generator BASEPOINT_TABLE projective", lambda: eng.mul_by_generator_batch(r, PROJECTIVE))
if lg <= 20:
    timed("mul_by_generator BASEPOINT_LOOKUP jacobian", lambda: eng.mul_by_generator_batch(r, JACOBIAN))
a = timed("multiply ct affine->affine (MJac)", lambda: eng.multiply_batch(pts, k, AFFINE, AFFINE))
timed("multiply ct affine->projective (Proj)", lambda: eng.multiply_batch(pts, k, AFFINE, PROJECTIVE))
b = timed("multiply_vartime affine->affine (MJac)", lambda: eng.multiply_vartime_batch(pts, k, AFFINE, AFFINE))
timed("multiply_vartime affine->projective (Proj)", lambda: eng.multiply_vartime_batch(pts, k, AFFINE, PROJECTIVE))
timed("multiply_vartime affine->jacobian (MJac)", lambda: eng.multiply_vartime_batch(pts, k, AFFINE, JACOBIAN))
assert torch.equal(a[0].view(torch.int64), b[0].view(torch.int64))
eng.close()
