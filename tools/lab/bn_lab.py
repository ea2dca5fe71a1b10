#!/usr/bin/env python3
"""A/B timing of k_batch_normalize<Jacobian> builds (tools/lab/build.sh <name> -DLAB_BN ...) at 2^24 points, on an
identity-free batch and on bench.py's batch (an identity every 1000); every build's output is compared in full with the
product library's.   usage: bn_lab.py lib1.so [lib2.so ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cheetah_amd import AFFINE, JACOBIAN, Engine

eng = Engine(0)
nb = 1 << 24
aff, _ = eng.mul_by_generator_batch(eng.synth_scalars(4, nb), AFFINE)
lam = eng.synth_fp6(4, nb)
l2 = eng.fp6_square_batch(lam)
l3 = eng.fp6_mul_batch(l2, lam)
affi = aff.view(torch.int64)
x = eng.fp6_mul_batch(affi[:, 0:6].contiguous(), l2)
y = eng.fp6_mul_batch(affi[:, 6:12].contiguous(), l3)
jac = torch.cat([x.view(torch.int64), y.view(torch.int64), lam.view(torch.int64)], dim=1).contiguous()
del x, y, l2, l3, lam, aff, affi
jac_id = jac.clone()
jac_id[torch.arange(0, nb, 1000, device="cuda"), 12:18] = 0
for name, batch in (("identity-free", jac), ("1-in-1000 identities", jac_id)):
    want_xy, want_inf = eng.batch_normalize(batch, JACOBIAN)
    torch.cuda.synchronize()
    for path in sys.argv[1:]:
        lib = C.CDLL(os.path.abspath(path))
        lib.lab_batch_normalize_jac.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        xy = torch.empty((nb, 12), dtype=torch.int64, device="cuda")
        inf = torch.empty((nb,), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.lab_batch_normalize_jac(batch.data_ptr(), xy.data_ptr(), inf.data_ptr(), nb, C.c_void_p(st))
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
            best = min(best, e0.elapsed_time(e1))
        ok = torch.equal(xy, want_xy.view(torch.int64)) and torch.equal(inf, want_inf)
        print("%-22s %-28s %.3f ms  %.2f G points/s  equal=%s" % (name, os.path.basename(path), best, nb / best / 1e6, ok), flush=True)
eng.close()
