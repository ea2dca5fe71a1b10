"""Edge encodings through every point-taking entry point, GPU against the oracle, bit-exact.

The reference's point types accept more than "a canonical point of the curve": an `Fp` is any u64 (limbs >= p are
legal, src/fp.rs:57), an identity is anything with z = 0 (`is_identity()`, src/curve/projective.rs:932-934,
src/curve/jacobian.rs:924-926) -- and its unchecked formulas do return (X, Y, 0) with arbitrary X, Y.  One batch holds
  * identities as (X, Y, 0) with random X, Y, as the canonical (0, 1, 0) / (1, 1, 0), and with z = (p, p, p, p, p, p),
  * points whose z is 1 written as (p + 1, p, p, p, p, p), affine identities written as (p .. p ; p + 1, p .. p),
between ordinary points, in waves that hold no scalar with bit 255 set (so that the vartime entry points' fixed-window
fast paths run) and in one wave that does.  This is the class of input that exposed the (X, Y, 0) gap of the vartime fast
paths."""
import numpy as np
import pytest

from cheetah_amd import synth

pytestmark = pytest.mark.gpu

AFFINE, PROJECTIVE, JACOBIAN = 0, 1, 2
P = 0xFFFFFFFF00000001
Q = synth.Q
N = 320


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cheetah_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint64:
        return torch.from_numpy(a.view(np.int64)).cuda().view(torch.uint64)
    return torch.from_numpy(a).cuda()


def host(t):
    import torch
    if t.dtype == torch.uint64:
        return t.view(torch.int64).cpu().numpy().view(np.uint64)
    return t.cpu().numpy()


def batch(eng, ora, form, seed):
    """(points in `form`, the same as affine with (0, 1) identities)"""
    aff, _ = eng.mul_by_generator_batch(synth.scalars(seed, N), AFFINE)
    aff = aff.copy()
    z = synth.fp6_elements(seed + 1, N).reshape(N, 6)
    x, y = np.ascontiguousarray(aff[:, :6]), np.ascontiguousarray(aff[:, 6:])
    if form == PROJECTIVE:
        pts = np.concatenate([ora.fp6_mul_batch(x, z).reshape(N, 6), ora.fp6_mul_batch(y, z).reshape(N, 6), z], axis=1)
        proper = [0] * 6 + [1] + [0] * 5 + [0] * 6
    else:
        z2 = ora.fp6_square_batch(z).reshape(N, 6)
        z3 = ora.fp6_mul_batch(z2, z).reshape(N, 6)
        pts = np.concatenate([ora.fp6_mul_batch(x, z2).reshape(N, 6), ora.fp6_mul_batch(y, z3).reshape(N, 6), z], axis=1)
        proper = [1] + [0] * 5 + [1] + [0] * 5 + [0] * 6
    pts = np.ascontiguousarray(pts)
    ident = np.array([0] * 6 + [1] + [0] * 5, np.uint64)
    for base in (64, 128, 256):                      # waves 1, 2 (no wide scalar) and 4 (one wide scalar, see scalars())
        pts[base + 5, 12:] = 0                       # (X, Y, 0), X and Y arbitrary
        pts[base + 6] = proper
        pts[base + 7, 12:] = P                       # z = (p, .., p) = 0, non-canonical
        pts[base + 9, :12] = np.concatenate([x[base + 9], y[base + 9]])
        pts[base + 9, 12:] = [P + 1, P, P, P, P, P]  # z = 1, non-canonical
        for i in (5, 6, 7):
            aff[base + i] = ident
    return pts, aff


def noncanonical_affine(aff):
    a = aff.copy()
    ident = (a[:, :6] == 0).all(axis=1) & (a[:, 6] == 1) & (a[:, 7:] == 0).all(axis=1)
    a[ident, :] = P
    a[ident, 6] = P + 1
    return a


def scalars(seed, per=1):
    sc = synth.scalars(seed, N * per)
    edge = [0, 1, Q - 1, Q, (1 << 255) - 1, 16, 15]
    for i, k in enumerate(edge):
        sc[70 * per + i] = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8)
    sc[(256 + 40) * per] = np.frombuffer(int((1 << 255) + 3).to_bytes(32, "little"), np.uint8)   # wave 4 takes the NAF kernels
    return sc


@pytest.mark.parametrize("form", [PROJECTIVE, JACOBIAN])
def test_ladders_on_edge_encodings(eng, ora, form):
    pts, aff = batch(eng, ora, form, 500 + form)
    affn = noncanonical_affine(aff)
    k = scalars(510 + form)
    K = dev(k)
    for in_form, p in ((form, pts), (AFFINE, affn)):
        Pd = dev(p)
        for out_form in (AFFINE, form):
            for vt in (False, True):
                want, winf = ora.multiply_batch(p, in_form, k, out_form, vartime=vt)
                got, inf = (eng.multiply_vartime_batch if vt else eng.multiply_batch)(Pd, K, in_form, out_form)
                assert np.array_equal(host(got), want) and np.array_equal(host(inf), winf), (in_form, out_form, vt)
        # multi-scalar: m = 2 over consecutive pairs, and a P + b G
        for vt in (False, True):
            want, winf = ora.multiply_many_batch(p, in_form, k, 2, AFFINE, vartime=vt)
            got, inf = (eng.multiply_many_vartime_batch if vt else eng.multiply_many_batch)(Pd, K, 2, in_form, AFFINE)
            assert np.array_equal(host(got), want) and np.array_equal(host(inf), winf), (in_form, "many", vt)
        k2 = scalars(520 + form, 2)
        for out_form in (AFFINE, form):
            want, winf = ora.multiply_many_batch(p, in_form, k2, 1, out_form, vartime=True, with_basepoint=True)
            got, inf = eng.multiply_double_with_basepoint_vartime_batch(Pd, dev(k2), in_form, out_form)
            assert np.array_equal(host(got), want) and np.array_equal(host(inf), winf), (in_form, out_form, "a P + b G")
        # subgroup check and cofactor clearing = the vartime ladder with q and h
        qb = np.tile(np.frombuffer(int(Q).to_bytes(32, "little"), np.uint8), (N, 1))
        want, winf = ora.multiply_batch(p, in_form, qb, PROJECTIVE if in_form == PROJECTIVE else JACOBIAN, vartime=True)
        assert np.array_equal(host(eng.is_torsion_free_batch(Pd, in_form)), winf)
        hb = np.tile(np.frombuffer(int(0x2150B48E071EF610049BC3F5D54304E4A).to_bytes(32, "little"), np.uint8), (N, 1))
        want, winf = ora.multiply_batch(p, in_form, hb, AFFINE, vartime=True)
        got, inf = eng.clear_cofactor_batch(Pd, in_form, AFFINE)
        assert np.array_equal(host(got), want) and np.array_equal(host(inf), winf), (in_form, "clear_cofactor")


@pytest.mark.parametrize("form", [PROJECTIVE, JACOBIAN])
def test_elementwise_on_edge_encodings(eng, ora, form):
    pts, aff = batch(eng, ora, form, 600 + form)
    affn = noncanonical_affine(aff)
    other, _ = batch(eng, ora, form, 700 + form)
    Pd, Od, Ad = dev(pts), dev(other), dev(affn)
    xy, inf = eng.batch_normalize(Pd, form)
    wxy, winf = ora.batch_normalize(pts, form == JACOBIAN)
    assert np.array_equal(host(xy), wxy) and np.array_equal(host(inf), winf)
    assert np.array_equal(wxy, aff)                   # ... and they are the points the batch was built from
    for f in (AFFINE, form):
        p = affn if f == AFFINE else pts
        assert np.array_equal(host(eng.is_on_curve_batch(dev(p), f)), ora.is_on_curve_batch(p, f))
        assert np.array_equal(host(eng.neg_batch(dev(p), f)), ora.neg_batch(p, f))
    assert np.array_equal(host(eng.eq_batch(Ad, dev(aff), AFFINE)), np.ones(N, np.uint8))      # two encodings, one point
    assert np.array_equal(host(eng.eq_batch(Pd, Od, form)), ora.eq_batch(pts, other, form))
    assert np.array_equal(host(eng.add_batch(Pd, Od, form)), ora.add_batch(pts, form, other, form))
    assert np.array_equal(host(eng.add_batch(Od, Ad, form, AFFINE)), ora.add_batch(other, form, affn, AFFINE))
    assert np.array_equal(host(eng.double_multi_batch(Pd, 3, form)), ora.double_multi_batch(pts, form, 3))
    # encodings of the normalised batch, identities included, and back
    for rec, enc, decode in ((97, eng.to_uncompressed_batch, eng.from_uncompressed_batch), (49, eng.to_compressed_batch, eng.from_compressed_batch)):
        e = enc(Ad, inf)
        assert np.array_equal(host(e), ora.encode_batch(affn, winf, rec == 49))
        dxy, dinf, some = decode(e, True)
        assert np.array_equal(host(dxy)[winf == 0], aff[winf == 0]) and np.array_equal(host(dinf), winf) and host(some).all()
    # BasePointTable::create from an identity written (X, Y, 0) / from a non-canonical z = 1
    for idx in (64 + 5, 64 + 9):
        t = eng.basepoint_table(pts[idx:idx + 1], form) if form == PROJECTIVE else eng.basepoint_table(affn[idx:idx + 1], AFFINE)
        want = ora.basepoint_table_create(pts[idx] if form == PROJECTIVE else affn[idx], form if form == PROJECTIVE else AFFINE)
        assert np.array_equal(t.read(), want[0])
        t.close()
