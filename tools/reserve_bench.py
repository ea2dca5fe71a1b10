import sys; sys.path.insert(0, "/root/repo")
import torch
from cheetah_amd import AFFINE, Engine
eng = Engine(0)
n = 1 << 22
r = eng.synth_scalars(5, n, 1 << 26); k = eng.synth_scalars(5, n, 0)
pts, _ = eng.mul_by_generator_batch(r, AFFINE)
for res in (0, 8, 0, 8):
    eng.set_reserved_cus(res)
    eng._use_torch_stream(); eng.timer_start()
    eng.multiply_batch(pts, k, AFFINE, AFFINE)
    ms = eng.timer_stop_ms()
    print("reserved %d CUs: %.3f ms  %.3f M/s" % (res, ms, n / ms / 1e3))
eng.close()
