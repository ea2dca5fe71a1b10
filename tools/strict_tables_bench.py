#!/usr/bin/env python3
"""cheetah_multiply_batch_dev with and without cheetah_ctx_set_point_independent_tables, alternating in one process
(HIP events, 2^lg affine points, affine results); outputs compared in full.   usage: strict_tables_bench.py [lg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cheetah_amd import AFFINE, Engine
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
eng = Engine(0)
eng.set_launch_rounds(16)
n = 1 << lg
pts, _ = eng.mul_by_generator_batch(eng.synth_scalars(5, n, 1 << 26), AFFINE)
k = eng.synth_scalars(5, n, 0)
ref = eng.multiply_batch(pts, k, AFFINE, AFFINE)
torch.cuda.synchronize()
for rep in range(3):
    for on in (False, True):
        eng.set_point_independent_tables(on)
        eng._use_torch_stream(); eng.timer_start()
        out = eng.multiply_batch(pts, k, AFFINE, AFFINE)
        ms = eng.timer_stop_ms()
        same = torch.equal(out[0].view(torch.int64), ref[0].view(torch.int64)) and torch.equal(out[1], ref[1])
        print("point-independent tables %-3s 2^%d: %9.3f ms  %7.3f M/s  equal=%s" % ("on" if on else "off", lg, ms, n / ms / 1e3, same), flush=True)
eng.set_point_independent_tables(False)
eng.close()
