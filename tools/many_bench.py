#!/usr/bin/env python3
"""HIP-event timings of the multi-scalar (Straus) kernels: multiply_double (m = 2) and multiply_many (m = 4),
constant time and variable time, affine in / affine out, 2^18 groups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import AFFINE, Engine
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
eng = Engine(0)
if len(sys.argv) > 2:
    eng.set_launch_rounds(int(sys.argv[2]))   # rounds of the resident grid per launch (library default 2)
for m in (2, 4):
    g = 1 << lg
    n = g * m
    r = eng.synth_scalars(7, n, 1 << 26)
    k = eng.synth_scalars(7, n, 0)
    pts, _ = eng.mul_by_generator_batch(r, AFFINE)
    for name, f in (("multiply_many ct", eng.multiply_many_batch), ("multiply_many vartime", eng.multiply_many_vartime_batch)):
        best = 1e30
        for _ in range(2):
            eng._use_torch_stream(); eng.timer_start()
            out = f(pts, k, m, AFFINE, AFFINE)
            best = min(best, eng.timer_stop_ms())
        print("%-24s m=%d 2^%d groups: %9.3f ms  %7.3f M groups/s  %7.3f M scalar mults/s" % (name, m, lg, best, g / best / 1e3, n / best / 1e3), flush=True)
k2 = eng.synth_scalars(8, 2 << lg, 0)
p1, _ = eng.mul_by_generator_batch(eng.synth_scalars(8, 1 << lg, 1 << 26), AFFINE)
for ref in (True, False):  # the reference's Straus NAF schedule, then the composed one (a P + b G + one addition)
    eng.set_reference_schedule(ref)
    best = 1e30
    for _ in range(3):
        eng._use_torch_stream(); eng.timer_start()
        out = eng.multiply_double_with_basepoint_vartime_batch(p1, k2, AFFINE, AFFINE)
        best = min(best, eng.timer_stop_ms())
    print("%-24s     2^%d groups: %9.3f ms  %7.3f M groups/s  (%s)" % ("double_with_basepoint_vt", lg, best, (1 << lg) / best / 1e3,
                                                                      "Straus NAF kernel" if ref else "composed"), flush=True)
eng.close()
