"""The HIP path against the INDEPENDENT leg directly (no oracle in between): the projective / Jacobian representatives
the GPU returns for multiply, multiply_vartime, BasePointTable::multiply and the complete additions equal the
compositions of the published closed forms on Python integers (tests/test_oracle_closed_forms_cpu.py).  Bar: bit-exact."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

AFFINE, PROJECTIVE, JACOBIAN = 0, 1, 2


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cheetah_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def test_gpu_representatives_are_the_closed_form_compositions(eng):
    import test_oracle_closed_forms_cpu as CF
    from oracle import model as M            # the big-integer model: test infrastructure
    rng = np.random.default_rng(41)
    g_xy, _ = eng.mul_by_generator_batch(np.frombuffer((1).to_bytes(32, "little"), np.uint8).reshape(1, 32).copy(), AFFINE)
    G = (CF._t(g_xy[0, :6]), CF._t(g_xy[0, 6:]))
    assert M.is_on_curve(G)
    base = M.scalar_mul(int(rng.integers(2, 2**62)), G)
    pts = np.array([list(base[0]) + list(base[1])], np.uint64)
    jac_add_mixed = lambda p, q: CF.hehcc_add(p, (q[0], q[1], M.ONE))
    trip = lambda a: [CF._t(a[0, :6]), CF._t(a[0, 6:12]), CF._t(a[0, 12:])]
    for k in (1, 0x0f0f0f0f0f, int.from_bytes(rng.bytes(32), "little") >> 1, M.Q - 1):
        kb = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8).reshape(1, 32).copy()
        for vt in (False, True):
            f = eng.multiply_vartime_batch if vt else eng.multiply_batch
            want = CF._ladder(base, k, vt, CF.hehcc_proj_double, CF.hehcc_proj_add_mixed)
            assert trip(f(pts, kb, AFFINE, PROJECTIVE)[0]) == [want[0], want[1], want[2]], ("projective", k, vt)
            want = CF._ladder(base, k, vt, CF.hehcc_double, jac_add_mixed)
            assert trip(f(pts, kb, AFFINE, JACOBIAN)[0]) == [want[0], want[1], want[2]], ("jacobian", k, vt)
    # the complete formulas on arbitrary field elements
    n = 40
    x1, y1, z1, x2, y2, z2 = (CF._rand_fp6(rng, n) for _ in range(6))
    z1[1] = rng.integers(1, M.P, 6, dtype=np.uint64)      # rows 0 and 1 of _rand_fp6 are edge values, the same in every operand:
    z1[0, 0] -= np.uint64(1)                               # keep lhs != rhs and z != 0 (the selections are tested elsewhere)
    p, q = CF._rows(x1, y1, z1), CF._rows(x2, y2, z2)
    pt = [(CF._t(x1[i]), CF._t(y1[i]), CF._t(z1[i])) for i in range(n)]
    qt = [(CF._t(x2[i]), CF._t(y2[i]), CF._t(z2[i])) for i in range(n)]
    CF._expect([CF.rcb_add(a, b) for a, b in zip(pt, qt)], eng.add_batch(p, q, PROJECTIVE), "GPU ProjectivePoint::add")
    CF._expect([CF.hehcc_proj_double(a) for a in pt], eng.double_batch(p, PROJECTIVE), "GPU ProjectivePoint::double")
    CF._expect([CF.hehcc_add(a, b) for a, b in zip(pt, qt)], eng.add_batch(p, q, JACOBIAN), "GPU JacobianPoint::add")
    CF._expect([CF.hehcc_double(a) for a in pt], eng.double_batch(p, JACOBIAN), "GPU JacobianPoint::double")


def test_gpu_affine_results_are_the_models_points(eng):
    """L1 against the model alone: the affine points of the widened comb, the ladders, a P + b G, clear_cofactor and
    batch_normalize equal the textbook affine group law on Python integers (oracle/model.py), for a handful of inputs."""
    import test_oracle_closed_forms_cpu as CF
    from oracle import model as M
    rng = np.random.default_rng(43)
    sc = lambda k: np.frombuffer(int(k).to_bytes(32, "little"), np.uint8).copy()
    ks = [1, 2, 0x7fff, int.from_bytes(rng.bytes(31), "little"), M.Q - 1, M.Q + 5, (1 << 255) - 19]
    kb = np.stack([sc(k) for k in ks])
    g_xy, _ = eng.mul_by_generator_batch(kb[:1].copy(), AFFINE)
    G = (CF._t(g_xy[0, :6]), CF._t(g_xy[0, 6:]))
    aff = lambda a, i: (CF._t(a[i, :6]), CF._t(a[i, 6:]))
    xy, inf = eng.mul_by_generator_batch(kb, AFFINE)
    for i, k in enumerate(ks):
        assert inf[i] == 0 and aff(xy, i) == M.scalar_mul(k, G), ("k G", hex(k))
    P = M.scalar_mul(int(rng.integers(2, 2**62)), G)
    pts = np.array([list(P[0]) + list(P[1])] * len(ks), np.uint64)
    for vt in (False, True):
        f = eng.multiply_vartime_batch if vt else eng.multiply_batch
        xy, inf = f(pts, kb, AFFINE, AFFINE)
        for i, k in enumerate(ks):
            assert inf[i] == 0 and aff(xy, i) == M.scalar_mul(k, P), ("k P", hex(k), vt)
    pairs = np.ascontiguousarray(np.stack([kb, kb[::-1]], axis=1).reshape(2 * len(ks), 32))
    xy, inf = eng.multiply_double_with_basepoint_vartime_batch(pts, pairs, AFFINE, AFFINE)
    for i in range(len(ks)):
        want = M.aff_add(M.scalar_mul(ks[i], P), M.scalar_mul(ks[len(ks) - 1 - i], G))
        assert (inf[i] == 1 and want is None) or (inf[i] == 0 and aff(xy, i) == want), ("a P + b G", i)
    h = 0x2150b48e071ef610049bc3f5d54304e4a
    xy, inf = eng.clear_cofactor_batch(pts[:1].copy(), AFFINE, AFFINE)
    assert inf[0] == 0 and aff(xy, 0) == M.scalar_mul(h, P)
    z = CF._t(rng.integers(1, M.P, 6, dtype=np.uint64))
    jac = np.array([list(CF._mul(P[0], z, z)) + list(CF._mul(P[1], z, z, z)) + list(z), [1] * 12 + [0] * 6], np.uint64)
    xy, inf = eng.batch_normalize(jac, JACOBIAN)
    assert inf.tolist() == [0, 1] and aff(xy, 0) == P and aff(xy, 1) == (M.ZERO, M.ONE)
