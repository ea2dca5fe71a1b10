import sys, time, torch
sys.path.insert(0, "/root/repo")
from cheetah_amd import AFFINE, JACOBIAN, Engine
eng = Engine(0)
n = 1 << 16
r = eng.synth_scalars(4, n)
jac, _ = eng.mul_by_generator_batch(r, JACOBIAN)
big = jac.repeat((1 << 24) // n, 1).contiguous()
for rep in range(3):
    eng._use_torch_stream(); eng.timer_start()
    xy, inf = eng.batch_normalize(big, JACOBIAN)
    ms = eng.timer_stop_ms()
    print("batch_normalize 2^24: %.3f ms  %.2f Gpts/s  %.1f GB/s" % (ms, (1 << 24) / ms / 1e6, (1 << 24) * 241 / ms / 1e6))
