#!/usr/bin/env python3
"""CPU side of BASELINE configs 1-4: the oracle's C restatement of the reference path (kind "port": the
reference is Rust and cannot be built in this pipeline), OpenMP static split over the box's host threads,
on bounded samples of the same seeded workloads.  Lives under tests/ because it runs the oracle (test infrastructure); pytest does not collect it.
Inputs that need curve points are produced by the oracle itself (no GPU needed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cheetah_amd import synth
from oracle import binding as ora

ora.build()
cores = ora.lib().ora_num_threads()
def best(f, reps=2):
    b = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b
print("host threads: %d" % cores)
n = 1 << 20
a, b = synth.fp6_elements(1, n, 0), synth.fp6_elements(1, n, n)
t = best(lambda: ora.fp6_mul_batch(a, b))
print("config 1  Fp6 mul, 2^20 pairs (all of it)           : %8.2f ms  %9.3f M/s" % (t * 1e3, n / t / 1e6))
m = 1 << 17
sc = synth.scalars(2, m)
ora.mul_by_generator_batch(sc[:64], 0)
t = best(lambda: ora.mul_by_generator_batch(sc, 0))
print("config 2  BASEPOINT_TABLE.multiply, 2^17 of 2^22    : %8.2f ms  %9.3f M/s" % (t * 1e3, m / t / 1e6))
pts, _ = ora.mul_by_generator_batch(sc, 0)
m3 = 1 << 15
k = synth.scalars(3, m3)
t = best(lambda: ora.multiply_batch(pts[:m3], 0, k, 1, vartime=True))
print("config 3  ProjectivePoint::multiply_vartime, 2^15   : %8.2f ms  %9.3f M/s" % (t * 1e3, m3 / t / 1e6))
jac, _ = ora.mul_by_generator_batch(sc, 2)                       # Jacobian (X, Y, Z) of k_i G
big = np.tile(jac, (8, 1))                                        # 2^20 Jacobian points
t1 = best(lambda: ora.batch_normalize(big, True, 0), reps=1)
print("config 4  batch_normalize, 2^20 Jacobian, ONE chain (the reference's shape, 1 thread): %8.2f ms  %9.3f M/s" % (t1 * 1e3, len(big) / t1 / 1e6))
t = best(lambda: ora.batch_normalize(big, True, len(big) // (cores * 4)))
print("config 4  batch_normalize, 2^20 Jacobian, %d chains over %d threads              : %8.2f ms  %9.3f M/s" % (cores * 4, cores, t * 1e3, len(big) / t / 1e6))
