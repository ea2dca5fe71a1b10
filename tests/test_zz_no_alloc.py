"""cheetah_multiply_batch_dev_noalloc (the entry point that allocates nothing: its window-table arena
is a file-scope device array) against cheetah_multiply_batch_dev, through the C ABI, bit for bit:
two calls in a row on the same GPU, the second with a smaller input (the arena keeps the first
call's contents).  Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest

from cheetah_amd import _capi, synth

pytestmark = pytest.mark.gpu

AFFINE, PROJECTIVE, JACOBIAN = 0, 1, 2


def _run(lib, h, fn, pts, in_form, ks, out_form):
    """One call through the C ABI on device buffers; returns (out, inf) on the host."""
    import torch
    n = ks.shape[0]
    w = 12 if out_form == AFFINE else 18
    out = torch.full((n, w), -1, dtype=torch.int64, device="cuda")
    inf = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]          # hipMemGetInfo: device memory free on this GPU
    rc = fn(h, pts.data_ptr(), in_form, ks.data_ptr(), out.data_ptr(), inf.data_ptr(), n, out_form)
    assert rc == 0, lib.cheetah_last_error(h)
    assert lib.cheetah_ctx_synchronize(h) == 0
    if fn is lib.cheetah_multiply_batch_dev_noalloc:
        # the property the entry point exists for: the call took no device memory (nothing else runs in this process
        # between the two readings; torch allocates nothing here, the outputs exist already)
        assert torch.cuda.mem_get_info()[0] == free_before, "cheetah_multiply_batch_dev_noalloc changed the free device memory"
    return out.cpu().numpy(), inf.cpu().numpy()


def test_noalloc_entry_point_matches_the_allocating_one():
    import torch
    assert torch.cuda.is_available()
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.cheetah_ctx_create((C.c_int * 1)(0), 1, C.byref(h)) == 0, lib.cheetah_last_error(None)
    try:
        # inputs: P_i = r_i G (device), k_i seeded; edge cases: an identity input, scalars with the top nibble > 7
        # (16-entry tables), zero and one; 40 000 > 128 workgroups x 256 lanes, so the capped grid strides
        for n, seed in ((40000, 71), (1000, 72)):
            r = torch.from_numpy(synth.scalars(seed, n, 1 << 20)).cuda()
            ks_h = synth.scalars(seed, n).copy()
            ks_h[3] = 0
            ks_h[4] = 0
            ks_h[4, 0] = 1
            ks_h[5] = 0xFF                      # 2^256 - 1: top digit 16
            ks_h[n - 1, 31] |= 0xF0
            ks = torch.from_numpy(ks_h).cuda()
            pts = torch.empty((n, 12), dtype=torch.int64, device="cuda")
            pinf = torch.empty((n,), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()               # the context runs on its own stream, torch on another
            assert lib.cheetah_mul_by_generator_batch_dev(h, r.data_ptr(), pts.data_ptr(), pinf.data_ptr(), n, AFFINE) == 0
            assert lib.cheetah_ctx_synchronize(h) == 0
            pts[7] = torch.tensor([0] * 6 + [1] + [0] * 5, dtype=torch.int64, device="cuda")   # the identity
            torch.cuda.synchronize()
            for in_form, out_form in ((AFFINE, AFFINE), (AFFINE, JACOBIAN), (AFFINE, PROJECTIVE)):
                want, winf = _run(lib, h, lib.cheetah_multiply_batch_dev, pts, in_form, ks, out_form)
                got, ginf = _run(lib, h, lib.cheetah_multiply_batch_dev_noalloc, pts, in_form, ks, out_form)
                assert (got == want).all() and (ginf == winf).all(), (n, in_form, out_form)
                assert (want != -1).any()
                if out_form == AFFINE:
                    assert winf[7] == 1 and winf[3] == 1      # identity input, zero scalar
        # sizes beyond the maximum are rejected with a clear error, nothing is launched
        big = _capi.load().cheetah_multiply_batch_dev_noalloc(h, pts.data_ptr(), AFFINE, ks.data_ptr(), pts.data_ptr(), pinf.data_ptr(),
                                                              (1 << 20) + 1, AFFINE)
        assert big == _capi.ERR_INVALID and b"CHEETAH_NOALLOC_MAX_N" in lib.cheetah_last_error(h)
        # n = 0 is a no-op
        assert lib.cheetah_multiply_batch_dev_noalloc(h, None, AFFINE, None, None, None, 0, AFFINE) == 0
    finally:
        lib.cheetah_ctx_destroy(h)


def test_noalloc_entry_point_under_hip_graph_capture():
    """The callers the entry point is for: its launches captured ONCE into a HIP graph (stream capture forbids
    hipMalloc / hipFree / synchronisation, and the call makes none), the graph replayed twice -- the second time with
    new scalars in the same buffers -- and both results equal to the allocating entry point's."""
    import torch
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.cheetah_ctx_create((C.c_int * 1)(0), 1, C.byref(h)) == 0, lib.cheetah_last_error(None)
    try:
        n = 30000
        r = torch.from_numpy(synth.scalars(75, n, 1 << 20)).cuda()
        ks = torch.from_numpy(synth.scalars(75, n)).cuda()
        pts = torch.empty((n, 12), dtype=torch.int64, device="cuda")
        pinf = torch.empty((n,), dtype=torch.uint8, device="cuda")
        want = torch.empty((n, 12), dtype=torch.int64, device="cuda")
        winf = torch.empty((n,), dtype=torch.uint8, device="cuda")
        out = torch.full((n, 12), -1, dtype=torch.int64, device="cuda")
        inf = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        assert lib.cheetah_mul_by_generator_batch_dev(h, r.data_ptr(), pts.data_ptr(), pinf.data_ptr(), n, AFFINE) == 0
        assert lib.cheetah_ctx_synchronize(h) == 0

        def reference():
            assert lib.cheetah_ctx_use_own_stream(h) == 0
            assert lib.cheetah_multiply_batch_dev(h, pts.data_ptr(), AFFINE, ks.data_ptr(), want.data_ptr(), winf.data_ptr(), n, AFFINE) == 0
            assert lib.cheetah_ctx_synchronize(h) == 0

        reference()
        s = torch.cuda.Stream()
        assert lib.cheetah_ctx_set_stream(h, C.c_void_p(s.cuda_stream)) == 0
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            rc = lib.cheetah_multiply_batch_dev_noalloc(h, pts.data_ptr(), AFFINE, ks.data_ptr(), out.data_ptr(), inf.data_ptr(), n, AFFINE)
        assert rc == 0, lib.cheetah_last_error(h)
        torch.cuda.synchronize()
        assert bool((out == -1).all()), "capture must record the launches, not run them"
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want) and torch.equal(inf, winf)
        ks.copy_(torch.from_numpy(synth.scalars(76, n)).cuda())
        torch.cuda.synchronize()
        reference()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want) and torch.equal(inf, winf)
        del g
    finally:
        lib.cheetah_ctx_destroy(h)
