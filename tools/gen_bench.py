#!/usr/bin/env python3
"""mul_by_generator (config 2: 2^22, constant time) HIP-event timing; CHEETAH_HIP_LIB selects a build variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import AFFINE, PROJECTIVE, Engine
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
eng = Engine(0)
n = 1 << lg
r = eng.synth_scalars(5, n, 1 << 26)
for form, name in ((AFFINE, "affine"), (PROJECTIVE, "projective")):
    best = 1e30
    for _ in range(3):
        eng._use_torch_stream(); eng.timer_start()
        out, inf = eng.mul_by_generator_batch(r, form)
        best = min(best, eng.timer_stop_ms())
    print("%s mul_by_generator %-10s 2^%d: %8.3f ms  %7.2f M/s  checksum %016x" % (os.environ.get("CHEETAH_HIP_LIB", "default").split("/")[-1], name, lg, best, n / best / 1e3,
          int(out.view(torch.int64).sum().item()) & 0xFFFFFFFFFFFFFFFF), flush=True)
eng.close()
