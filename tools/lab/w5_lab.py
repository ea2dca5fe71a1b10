#!/usr/bin/env python3
"""VERDICT r02 item 1b measured: AffinePoint::multiply, affine results, by signed 5-bit windows (52 additions + 260
doublings, 16-entry table) against the product's radix-16 ladder (64 + 256, 8 entries), constant-time (scan) and
indexed (multiply_vartime's fast path) lookups, same launch shape, same box; outputs compared in full with the product
library's.  Edge items (0, 1, q - 1, 2^255 - 1, identity points, a ragged tail) included.
    tools/lab/build.sh w5 -DLAB_W5 ;  usage: w5_lab.py <lg> libw5.so"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from cheetah_amd import AFFINE, Engine, synth
import lab_common as L

lg = int(sys.argv[1])
lib = L.load(sys.argv[2])
eng = Engine(0)
n = (1 << lg) - 37
pts, _ = eng.mul_by_generator_batch(eng.synth_scalars(5, n, 1 << 26), AFFINE)
kh = eng.synth_scalars(5, n, 0).cpu().numpy().copy()
kh[0] = 0
kh[1] = 0; kh[1, 0] = 1
kh[2] = np.frombuffer((synth.Q - 1).to_bytes(32, "little"), np.uint8)
kh[3] = 0xFF; kh[3, 31] = 0x7F
kh[n - 1] = 0
k = torch.from_numpy(kh).cuda()
pts[5] = torch.tensor([0] * 6 + [1] + [0] * 5, dtype=torch.int64, device="cuda").view(pts.dtype)
pts[300] = pts[5]
torch.cuda.synchronize()
tab = L.tables(lib, eng)
lib.lab_ct_scratch_words.restype = C.c_size_t
lib.lab_w5_consts.argtypes = [C.c_void_p, C.c_void_p]
lib.lab_multiply_w.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p, C.c_void_p]
consts = torch.zeros((24,), dtype=torch.int64, device="cuda")
assert lib.lab_w5_consts(tab, consts.data_ptr()) == 0
scratch = torch.empty((lib.lab_ct_scratch_words(),), dtype=torch.int64, device="cuda")
want, winf = eng.multiply_batch(pts, k, AFFINE, AFFINE)
torch.cuda.synchronize()
for direct in (0, 1):
    for w in (4, 5, 4, 5):
        out = torch.full((n, 12), -1, dtype=torch.int64, device="cuda")
        inf = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ms = L.timed(lambda: lib.lab_multiply_w(w, direct, tab, consts.data_ptr(), pts.data_ptr(), k.data_ptr(), out.data_ptr(), inf.data_ptr(), n,
                                                scratch.data_ptr(), st))
        ok = torch.equal(out, want.view(torch.int64)) and torch.equal(inf, winf)
        print("%-8s w=%d  2^%d: %9.3f ms  %7.3f M/s  equal=%s" % ("indexed" if direct else "ct scan", w, lg, ms, n / ms / 1e3, ok), flush=True)
# the co-Z upper bound: the product's ladder minus 2 S per window (results wrong on purpose, see lab.hip)
for rep in range(2):
    out = torch.full((n, 12), -1, dtype=torch.int64, device="cuda")
    inf = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = L.timed(lambda: lib.lab_multiply_w(40, 0, tab, consts.data_ptr(), pts.data_ptr(), k.data_ptr(), out.data_ptr(), inf.data_ptr(), n,
                                            scratch.data_ptr(), st))
    print("ct scan  w=4 minus 2 S per window (co-Z upper bound, not a multiplication)  2^%d: %9.3f ms  %7.3f M/s" % (lg, ms, n / ms / 1e3), flush=True)
eng.close()
