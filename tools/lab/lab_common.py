"""Shared plumbing of the kernel-lab harnesses: loads a lab build, builds its DeviceTables from the product's tables."""
import ctypes as C
import os

import numpy as np
import torch


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    lib.lab_build_tables.restype = C.c_void_p
    lib.lab_build_tables.argtypes = [C.c_void_p, C.c_void_p]
    return lib


def tables(lib, eng):
    shift = eng.read_table(1)[0:1]            # S
    gen = eng.read_table(2)[0:1]              # BASEPOINT_LOOKUP[0] = G
    g = torch.from_numpy(np.ascontiguousarray(gen).view(np.int64)).cuda()
    s = torch.from_numpy(np.ascontiguousarray(shift).view(np.int64)).cuda()
    torch.cuda.synchronize()
    tab = lib.lab_build_tables(g.data_ptr(), s.data_ptr())
    assert tab, "lab_build_tables failed"
    return C.c_void_p(tab)


def timed(launch, reps=3):
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = launch()
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        best = min(best, e0.elapsed_time(e1))
    return best
