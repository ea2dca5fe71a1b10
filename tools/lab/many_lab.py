#!/usr/bin/env python3
"""A/B timing of k_multiply_many_ct builds (tools/lab/build.sh <name> -DLAB_MANY -DCH_MANY_WAVES=..):
multiply_many constant time and its indexed-lookup (vartime, affine results) form, m = 2 and 4, 2^lg groups, affine
in / out; outputs compared in full with the product library's.   usage: many_lab.py <lg> lib1.so [lib2.so ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from cheetah_amd import AFFINE, Engine
import lab_common as L

lg = int(sys.argv[1])
eng = Engine(0)
libs = [(os.path.basename(p), L.load(p)) for p in sys.argv[2:]]
for m in (2, 4):
    g = 1 << lg
    n = g * m
    pts, _ = eng.mul_by_generator_batch(eng.synth_scalars(7, n, 1 << 26), AFFINE)
    k = eng.synth_scalars(7, n, 0)
    want = {0: eng.multiply_many_batch(pts, k, m, AFFINE, AFFINE), 1: eng.multiply_many_vartime_batch(pts, k, m, AFFINE, AFFINE)}
    torch.cuda.synchronize()
    for name, lib in libs:
        lib.lab_many_scratch_words.restype = C.c_size_t
        lib.lab_multiply_many.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        tab = L.tables(lib, eng)
        scratch = torch.empty((lib.lab_many_scratch_words(m),), dtype=torch.int64, device="cuda")
        for vt in (0, 1):
            out = torch.empty((g, 12), dtype=torch.int64, device="cuda")
            inf = torch.empty((g,), dtype=torch.uint8, device="cuda")
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            ms = L.timed(lambda: lib.lab_multiply_many(tab, pts.data_ptr(), k.data_ptr(), out.data_ptr(), inf.data_ptr(), g, m,
                                                       scratch.data_ptr(), vt, st))
            ok = torch.equal(out, want[vt][0].view(torch.int64)) and torch.equal(inf, want[vt][1])
            print("%-26s m=%d %-8s 2^%d groups: %8.3f ms  %7.3f M groups/s  equal=%s" % (name, m, "indexed" if vt else "ct", lg, ms, g / ms / 1e3, ok), flush=True)
        del scratch
eng.close()
