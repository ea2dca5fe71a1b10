"""GPU parity of the operations added in round 3 around the hot path, through the C ABI:
  * Fp::sqrt (src/fp.rs:199-229) and Fp::exp / exp_vartime (:304-333) as operations of cheetah_fp_op_batch,
  * Fp6::exp / exp_vartime (src/fp6.rs:1053-1086) as cheetah_fp6_exp_batch,
  * the seeded counterpart of {Affine,Projective,Jacobian}Point::random (src/curve/projective.rs:228-248,
    jacobian.rs:247-267, affine.rs:243-245) as cheetah_random_points_dev.
Checked against the oracle and against what the reference's own tests hold: sqrt(0) = 0, sqrt(1) = 1, 7 is a
non-square (src/fp.rs:899-910); a^(p^6 - 2) = a^-1 (src/fp6.rs:1760-1789); a^p = frobenius(a) (:1792-1802).
Bar: bit-exact."""
import numpy as np
import pytest

from cheetah_amd import synth

pytestmark = pytest.mark.gpu

AFFINE, PROJECTIVE, JACOBIAN = 0, 1, 2
P = 0xFFFFFFFF00000001
H = 0x2150b48e071ef610049bc3f5d54304e4a          # cofactor, src/curve/projective.rs:873-878


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


def test_fp_sqrt_and_exp(eng, ora):
    rng = np.random.default_rng(81)
    x = rng.integers(0, P, 3000, dtype=np.uint64)
    sq = np.array([pow(int(v), 2, P) for v in x[:1000]], np.uint64)
    vals = np.concatenate([np.array([0, 1, 7, P, P + 1, P - 1, 2**64 - 1, 4, 2**32], np.uint64), sq, x])
    want, wsome = ora.fp_sqrt(vals)
    assert wsome[0] == 1 and want[0] == 0 and wsome[1] == 1 and want[1] == 1 and wsome[2] == 0     # src/fp.rs:899-910
    assert wsome[9:1009].all() and 0.4 < wsome[1009:].mean() < 0.6
    for arr in (vals, dev(vals)):
        got, gsome = eng.fp_op_batch("sqrt", arr)
        got, gsome = (host(got), host(gsome)) if not isinstance(got, np.ndarray) else (got, gsome)
        assert (got == want).all() and (gsome == wsome).all()
        ok = gsome == 1
        assert all(pow(int(r), 2, P) == int(v) % P for r, v in zip(got[ok], vals[ok]))
    for e in (0, 1, 2, 0x7fffffff, P - 2, P - 1, 2**64 - 1, 0x123456789abcdef):
        want = np.array([pow(int(v), e, P) for v in vals], np.uint64)
        assert all(ora.lib().ora_fp_exp(int(v), e) == int(w) for v, w in zip(vals[:40], want[:40]))
        assert (eng.fp_op_batch("exp", vals, scalar=e) == want).all()
        assert (host(eng.fp_op_batch("exp", dev(vals), scalar=e)) == want).all()
    inv, _ = eng.fp_op_batch("invert", vals)
    assert (eng.fp_op_batch("exp", vals, scalar=P - 2) == inv).all()                             # src/fp.rs:914-934


def test_fp6_exp(eng, ora):
    n = 700
    a = synth.fp6_elements(82, n).copy()
    a[0] = 0
    a[1] = [1, 0, 0, 0, 0, 0]
    a[2] = [P, P, P, P, P, P]                  # non-canonical zero
    a[3] = [2**64 - 1] * 6
    rnd = int.from_bytes(np.random.default_rng(83).bytes(48), "little")
    for power in (0, 1, 2, P, P**6 - 2, rnd, 2**384 - 1):
        want = ora.fp6_exp_batch(a, power)
        assert (eng.fp6_exp_batch(a, power) == want).all(), hex(power)
        assert (host(eng.fp6_exp_batch(dev(a), power)) == want).all(), hex(power)
    # what the reference's tests check: a^(p^6 - 2) is the inverse, a^p the Frobenius map
    inv, some = eng.fp6_invert_batch(a)
    e = eng.fp6_exp_batch(a, P**6 - 2)
    assert (e[some == 1] == inv[some == 1]).all() and (e[some == 0] == 0).all()
    assert (eng.fp6_exp_batch(a, P) == eng.fp6_op_batch("frobenius", a)).all()


def _random_candidates_model(ora, seed, start, n):
    """the candidate step of cheetah_random_points_dev restated with numpy + the oracle's Fp6 operations"""
    out = np.zeros((n, 12), np.uint64)
    out[:, 6] = 1
    tries = np.zeros(n, np.int64)
    todo = np.arange(n)
    for a in range(36):
        if len(todo) == 0:
            break
        base = (np.uint64(start) + todo.astype(np.uint64)) * np.uint64(256) + np.uint64(7 * a)
        x = np.stack([synth.splitmix64(seed, base + np.uint64(l)) % np.uint64(P) for l in range(6)], axis=1)
        flip = (synth.splitmix64(seed, base + np.uint64(6)) & np.uint64(1)).astype(bool)
        x3 = ora.fp6_mul_batch(ora.fp6_square_batch(x), x)
        rhs = np.stack([ora.fp6_binop("ora_fp6_add", x3[i], x[i]) for i in range(len(x))])
        rhs[:, 0] = (rhs[:, 0] % np.uint64(P) + np.uint64(395)) % np.uint64(P)
        rhs[:, 1] = (rhs[:, 1] % np.uint64(P) + np.uint64(1)) % np.uint64(P)
        y, ok = ora.fp6_sqrt_batch(rhs)
        y = np.where(flip[:, None], (np.uint64(P) - y % np.uint64(P)) % np.uint64(P), y % np.uint64(P))
        hit = ok == 1
        out[todo[hit], :6] = x[hit]
        out[todo[hit], 6:] = y[hit]
        tries[todo[hit]] = a + 1
        todo = todo[~hit]
    return out, tries


def test_random_points_are_the_reference_construction(eng, ora):
    seed, n, start = 91, 600, 5
    cand, tries = _random_candidates_model(ora, seed, start, n)
    assert tries.min() >= 1 and tries.max() >= 4 and 1.6 < tries.mean() < 2.4        # half of all x have a point over them
    assert ora.is_on_curve_batch(cand, 0).all()
    hb = np.repeat(np.frombuffer(H.to_bytes(32, "little"), np.uint8).reshape(1, 32), n, axis=0)
    for form in (AFFINE, PROJECTIVE, JACOBIAN):
        want, winf = ora.multiply_batch(cand, 0, hb, form, vartime=True)            # clear_cofactor = multiply_vartime(h)
        got, ginf = eng.random_points(seed, n, start, form)
        assert (host(got) == want).all() and (host(ginf) == winf).all() and not winf.any(), form
    xy, inf = eng.random_points(seed, n, start, AFFINE)
    assert host(eng.is_on_curve_batch(xy, AFFINE)).all() and host(eng.is_torsion_free_batch(xy, AFFINE)).all()
    # the stream is addressed by element: a later window of it is the same points
    xy2, _ = eng.random_points(seed, 100, start + 250, AFFINE)
    assert (host(xy2) == host(xy)[250:350]).all()
    # the host-pointer entry point fills numpy arrays with the same stream
    for form in (AFFINE, PROJECTIVE):
        hxy, hinf = eng.random_points(seed, 300, start, form, host=True)
        dxy, dinf = eng.random_points(seed, 300, start, form)
        assert isinstance(hxy, np.ndarray) and (hxy == host(dxy)).all() and (hinf == host(dinf)).all()
    other, _ = eng.random_points(seed + 1, 100, start, AFFINE)
    assert (host(other) != host(xy)[:100]).any(axis=1).all()


def test_random_points_large_batch_properties(eng):
    """2^19 + 3 points (two sub-batches): all on the curve, all in the prime-order subgroup, all distinct x."""
    import torch
    n = (1 << 19) + 3
    xy, inf = eng.random_points(92, n, 0, AFFINE)
    assert int(inf.sum()) == 0
    assert bool(eng.is_on_curve_batch(xy, AFFINE).all()) and bool(eng.is_torsion_free_batch(xy, AFFINE).all())
    x0 = xy.view(torch.int64)[:, 0]
    assert torch.unique(x0).numel() == n
