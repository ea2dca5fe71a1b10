#!/usr/bin/env python3
"""One launch of each hot-path kernel at a profiling-friendly size, for
`rocprofv3 --kernel-trace --stats` and `--pmc` passes (see profiles/README.md).

    python3 tools/prof_driver.py [log2_ladders] [log2_fp6]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cheetah_amd import AFFINE, JACOBIAN, Engine  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
lf = int(sys.argv[2]) if len(sys.argv) > 2 else 24
eng = Engine(0)
n = 1 << lg
r = eng.synth_scalars(5, n, 1 << 26)
pts, inf = eng.mul_by_generator_batch(r, AFFINE)          # k_mul_by_generator<0>
k = eng.synth_scalars(5, n, 0)
for _ in range(2):
    out, oinf = eng.multiply_batch(pts, k, AFFINE, AFFINE)         # k_multiply_ct<MJac,0,0>
vt, vinf = eng.multiply_vartime_batch(pts, k, AFFINE, AFFINE)  # k_multiply_vartime<MJac,0,0>
from cheetah_amd import PROJECTIVE  # noqa: E402
pr, pinf = eng.multiply_vartime_batch(pts, k, AFFINE, PROJECTIVE)   # k_multiply_vartime_pool<Proj,0,1>
jv, jvinf = eng.multiply_vartime_batch(pts, k, AFFINE, JACOBIAN)    # k_multiply_vartime_pool<MJac,0,2>
mm, minf = eng.multiply_many_batch(pts, k, 2, AFFINE, AFFINE)      # k_multiply_many_ct<MJac,0,0>
jac, jinf = eng.multiply_batch(pts, k, AFFINE, JACOBIAN)
nb = min(1 << 24, n * 16)
big = jac.repeat(nb // n, 1).contiguous()
xy, binf = eng.batch_normalize(big, JACOBIAN)              # k_batch_normalize<1>
nf = 1 << lf
a, b = eng.synth_fp6(1, nf, 0), eng.synth_fp6(1, nf, nf)
for _ in range(3):
    c = eng.fp6_mul_batch(a, b)                             # k_fp6_elementwise<0>
a2, b2 = eng.synth_fp6(1, 1 << 20, 0), eng.synth_fp6(1, 1 << 20, 1 << 20)
for _ in range(3):
    c2 = eng.fp6_mul_batch(a2, b2)
torch.cuda.synchronize()
assert torch.equal(out.view(torch.int64), vt.view(torch.int64))
print("prof_driver ok: ladders 2^%d, fp6 2^%d" % (lg, lf))
eng.close()
