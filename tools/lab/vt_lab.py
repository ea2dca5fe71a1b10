#!/usr/bin/env python3
"""k_multiply_vartime (one item per lane, wave parking) against k_multiply_vartime_pool (SLOTS x 64 items per wave) at
2^lg multiplications, affine in, the reference's projective / Jacobian representative out; both compared in full with
the product library's output.  Edge items (zero / one / q - 1 / wide scalars, identity points, a ragged tail) included.
    usage: vt_lab.py <lg> lib.so [waves_per_simd]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE, Engine, synth
import lab_common as L

lg = int(sys.argv[1])
lib = L.load(sys.argv[2])
bpc = int(sys.argv[3]) if len(sys.argv) > 3 else 4
eng = Engine(0)
n = (1 << lg) - 37                       # ragged
pts, _ = eng.mul_by_generator_batch(eng.synth_scalars(5, n, 1 << 26), AFFINE)
k = eng.synth_scalars(5, n, 0)
kh = k.cpu().numpy().copy()
kh[0] = 0
kh[1] = 0; kh[1, 0] = 1
kh[2] = np.frombuffer((synth.Q - 1).to_bytes(32, "little"), np.uint8)
kh[3] = 0xFF
kh[70, 31] |= 0x80
kh[n - 1] = 0
k = torch.from_numpy(kh).cuda()
pts[5] = torch.tensor([0] * 6 + [1] + [0] * 5, dtype=torch.int64, device="cuda").view(pts.dtype)
pts[300] = pts[5]
torch.cuda.synchronize()
tab = L.tables(lib, eng)
lib.lab_scratch_words_per_block.restype = C.c_size_t
lib.lab_pool_words_per_block.restype = C.c_size_t
lib.lab_pool_words_per_block.argtypes = [C.c_int]
sig = [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.lab_multiply_vartime.argtypes = sig
lib.lab_multiply_vartime_pool.argtypes = sig
ncu = lib.lab_num_cus()
for proj, form in ((1, PROJECTIVE), (0, JACOBIAN)):
    want, winf = eng.multiply_vartime_batch(pts, k, AFFINE, form)
    torch.cuda.synchronize()
    for name, f, words in (("parked (1 item / lane)", lib.lab_multiply_vartime, lib.lab_scratch_words_per_block()),
                           ("pool", lib.lab_multiply_vartime_pool, lib.lab_pool_words_per_block(proj))):
        scratch = torch.empty((ncu * bpc * words,), dtype=torch.int64, device="cuda")
        out = torch.full((n, 18), -1, dtype=torch.int64, device="cuda")
        inf = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ms = L.timed(lambda: f(tab, pts.data_ptr(), k.data_ptr(), out.data_ptr(), inf.data_ptr(), n, scratch.data_ptr(), bpc, proj, st))
        ok = torch.equal(out, want.view(torch.int64)) and torch.equal(inf, winf)
        bad = int((out != want.view(torch.int64)).any(dim=1).sum())
        if name == "pool" and hasattr(lib, "lab_read_counters"):
            c = (C.c_ulonglong * 4)()
            lib.lab_read_counters(c, 1)
            waves = -(-n // 128) * 3      # waves x timed repetitions (3)
            print("   phases per wave: %.1f doubling (%.1f %% of lanes), %.1f addition (%.1f %%)" % (
                c[0] / waves * 1.0, 100.0 * c[2] / max(1, 64 * c[0]), c[1] / waves * 1.0, 100.0 * c[3] / max(1, 64 * c[1])))
        print("%-10s %-24s 2^%d: %9.3f ms  %7.3f M/s  equal=%s (rows differing: %d)" % ("projective" if proj else "jacobian", name, lg, ms, n / ms / 1e3, ok, bad), flush=True)
        del scratch
eng.close()
