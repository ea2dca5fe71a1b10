"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle (oracle/cheetah_oracle.c), the committed golden fixtures and the
reference's own known-answer vectors.  Bar: bit-exact on canonical limbs."""
import json
import os

import numpy as np
import pytest

from cheetah_amd import synth

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001
Q = synth.Q
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def take(t, idx):
    """Rows idx of a device tensor (uint64 tensors cannot be indexed on the device: go through int64)."""
    import torch
    if t.dtype == torch.uint64:
        return t.view(torch.int64)[idx].view(torch.uint64)
    return t[idx]


EDGE_FP = [0, 1, 2, 7, P - 1, P, P + 1, 2**64 - 1, 2**32, 2**32 - 1, 2**32 + 1, 2**63, 2**63 - 1, P - 2**32, 0xFFFFFFFF00000000]


def edge_fp6(n, seed):
    """Seeded Fp6 elements with edge / non-canonical limbs mixed in."""
    a = synth.fp6_elements(seed, n).copy()
    rng = np.random.default_rng(seed)
    for i in range(min(n, 4 * len(EDGE_FP))):
        a[i, rng.integers(0, 6)] = EDGE_FP[i % len(EDGE_FP)]
    if n > 70:
        a[64] = [2**64 - 1] * 6
        a[65] = [P - 1] * 6
        a[66] = [0] * 6
        a[67] = [P] * 6  # non-canonical zero
        a[68] = [1, 0, 0, 0, 0, 0]
    return a


# ------------------------------------------------------- device functions ----

def test_fp_device_functions(eng):
    rng = np.random.default_rng(3)
    vals = EDGE_FP + [int(v) for v in rng.integers(0, 2**64, 2000, dtype=np.uint64)]
    a = np.array(vals, np.uint64)
    b = np.array(vals[::-1], np.uint64)
    b[: len(EDGE_FP) ** 1] = a[: len(EDGE_FP)][::-1]
    # all edge pairs
    ea, eb = np.meshgrid(np.array(EDGE_FP, np.uint64), np.array(EDGE_FP, np.uint64))
    a = np.concatenate([a, ea.ravel()])
    b = np.concatenate([b, eb.ravel()])
    A, B = dev(a.reshape(-1, 1)), dev(b.reshape(-1, 1))
    ai = [int(v) for v in a]
    bi = [int(v) for v in b]
    got = host(eng.debug_op(0, A, B, 1)).ravel()
    assert [int(g) % P for g in got] == [(x + y) % P for x, y in zip(ai, bi)]
    got = host(eng.debug_op(1, A, B, 1)).ravel()
    assert [int(g) % P for g in got] == [(x - y) % P for x, y in zip(ai, bi)]
    got = host(eng.debug_op(2, A, B, 1)).ravel()
    assert [int(g) % P for g in got] == [(x * y) % P for x, y in zip(ai, bi)]
    got = host(eng.debug_op(3, A, None, 1)).ravel()
    assert [int(g) % P for g in got] == [(7 * x) % P for x in ai]
    got = host(eng.debug_op(4, A, None, 1)).ravel()
    assert [int(g) % P for g in got] == [pow(x % P, P - 2, P) for x in ai]  # inverse = exp(p-2), src/fp.rs:914-934


def test_fp6_device_functions(eng, ora):
    a, b = edge_fp6(300, 31), edge_fp6(300, 32)[::-1].copy()
    A, B = dev(a), dev(b)
    L = ora.lib()
    for op, name, binary in ((16, "ora_fp6_add", True), (17, "ora_fp6_sub", True), (18, "ora_mul_by_3b", False),
                             (19, "ora_fp6_frobenius", False)):
        got = host(eng.debug_op(op, A, B if binary else None, 6)) % np.uint64(P)
        for i in range(len(a)):
            want = ora.canon(ora.fp6_binop(name, a[i], b[i]) if binary else ora.fp6_unop(name, a[i]))
            assert (got[i] == want).all(), (name, i)


CARRY_EDGE = sorted(set(EDGE_FP + [
    2**64 - 2, 2**64 - 2**32, 2**64 - 2**32 - 1, 2**64 - 2**32 + 2, 2**33 - 2, 2**33 - 1, 2**33, 2**32 - 2,
    0x00000001FFFFFFFF, 0xFFFFFFFE00000001, 0xFFFFFFFE00000002, 0xFFFFFFFEFFFFFFFF, 0x7FFFFFFF80000000,
    0x7FFFFFFF80000001, 0x8000000000000001, 0x00000000FFFFFFFE, 0xFFFFFFFF00000002, 0xFFFFFFFFFFFFFFFE - 2**32]))


def test_fp6_linear_and_mul_carry_edges(eng):
    """Every pair of carry-edge limbs through the carry-mask forms of Fp6 add / sub (fp_add3 / fp_sub3: sums and
    differences whose low word ends at 0 or 2^32 - 1, double folds) and all-edge operands through the
    accumulate-and-reduce of Fp6 mul / square, against Python integers."""
    E = CARRY_EDGE
    pairs = [(x, y) for x in E for y in E]
    while len(pairs) % 6:
        pairs.append((E[0], E[-1]))
    a = np.array([p[0] for p in pairs], np.uint64).reshape(-1, 6)
    b = np.array([p[1] for p in pairs], np.uint64).reshape(-1, 6)
    A, B = dev(a), dev(b)
    ai = [[int(v) for v in r] for r in a]
    bi = [[int(v) for v in r] for r in b]
    got = host(eng.debug_op(16, A, B, 6))
    assert [[int(g) % P for g in r] for r in got] == [[(x + y) % P for x, y in zip(ra, rb)] for ra, rb in zip(ai, bi)]
    got = host(eng.debug_op(17, A, B, 6))
    assert [[int(g) % P for g in r] for r in got] == [[(x - y) % P for x, y in zip(ra, rb)] for ra, rb in zip(ai, bi)]

    def mul_model(x, y):
        c = [0] * 11
        for i in range(6):
            for j in range(6):
                c[i + j] += x[i] * y[j]
        return [(c[k] + 7 * (c[k + 6] if k < 5 else 0)) % P for k in range(6)]

    rng = np.random.default_rng(77)
    n = 6000
    x = np.array(E, np.uint64)[rng.integers(0, len(E), (n, 6))]
    y = np.array(E, np.uint64)[rng.integers(0, len(E), (n, 6))]
    x[0], y[0] = [2**64 - 1] * 6, [2**64 - 1] * 6
    x[1], y[1] = [2**64 - 1] * 6, [P - 1] * 6
    got = eng.fp6_mul_batch(x, y)
    assert [list(map(int, r)) for r in got] == [mul_model(list(map(int, r)), list(map(int, s))) for r, s in zip(x, y)]
    got = eng.fp6_square_batch(x)
    assert [list(map(int, r)) for r in got] == [mul_model(list(map(int, r)), list(map(int, r))) for r in x]


def _random_points(ora, seed, n, form):
    """k_i * G as affine, projective or Jacobian (the latter two scaled by a random lambda)."""
    sc = synth.scalars(seed, n)
    xy, inf = ora.mul_by_generator_batch(sc, 0)
    if form == "affine":
        return xy
    lam = synth.fp6_elements(seed + 1000, n)
    out = np.zeros((n, 18), np.uint64)
    for i in range(n):
        if form == "jac":  # (x l^2, y l^3, l), the construction of src/curve/jacobian.rs:1383-1387
            l2 = ora.fp6_unop("ora_fp6_square", lam[i])
            out[i, 0:6] = ora.fp6_binop("ora_fp6_mul", xy[i, 0:6], l2)
            out[i, 6:12] = ora.fp6_binop("ora_fp6_mul", xy[i, 6:12], ora.fp6_binop("ora_fp6_mul", l2, lam[i]))
        else:
            out[i, 0:6] = ora.fp6_binop("ora_fp6_mul", xy[i, 0:6], lam[i])
            out[i, 6:12] = ora.fp6_binop("ora_fp6_mul", xy[i, 6:12], lam[i])
        out[i, 12:18] = lam[i]
    return out


def _to_form(ora, aff, form, seed):
    """Given affine points ((0, 1) = identity) -> projective (x l, y l, l) / Jacobian (x l^2, y l^3, l) representatives."""
    n = len(aff)
    lam = synth.fp6_elements(seed, n).reshape(n, 6)
    x, y = np.ascontiguousarray(aff[:, :6]), np.ascontiguousarray(aff[:, 6:])
    if form == "jac":
        l2 = ora.fp6_square_batch(lam).reshape(n, 6)
        out = np.concatenate([ora.fp6_mul_batch(x, l2).reshape(n, 6), ora.fp6_mul_batch(y, ora.fp6_mul_batch(l2, lam).reshape(n, 6)).reshape(n, 6), lam], axis=1)
    else:
        out = np.concatenate([ora.fp6_mul_batch(x, lam).reshape(n, 6), ora.fp6_mul_batch(y, lam).reshape(n, 6), lam], axis=1)
    ident = (aff[:, :6] == 0).all(axis=1) & (aff[:, 6] == 1) & (aff[:, 7:] == 0).all(axis=1)
    out[ident, 12:] = 0
    return np.ascontiguousarray(out)


def test_point_formulas_projective_parity(eng, ora):
    """double_unchecked / add / add_mixed_unchecked give the reference's own
    (X:Y:Z) representative (parity level L2), src/curve/projective.rs:400-610."""
    L = ora.lib()
    n = 40
    p = _random_points(ora, 41, n, "proj")
    q = _random_points(ora, 43, n, "proj")
    qa = _random_points(ora, 45, n, "affine")
    Pd, Qd, QAd = dev(p), dev(q), dev(qa)
    got_dbl = host(eng.debug_op(32, Pd, None, 18))
    got_add = host(eng.debug_op(33, Pd, Qd, 18))
    got_mix = host(eng.debug_op(34, Pd, QAd, 18))
    out = np.zeros(18, np.uint64)
    for i in range(n):
        L.ora_proj_double_unchecked(np.ascontiguousarray(p[i]), out)
        assert (got_dbl[i] == out % np.uint64(P)).all(), i
        L.ora_proj_add(np.ascontiguousarray(p[i]), np.ascontiguousarray(q[i]), out)
        assert (got_add[i] == out % np.uint64(P)).all(), i
        L.ora_proj_add_mixed_unchecked(np.ascontiguousarray(p[i]), np.ascontiguousarray(qa[i]), 0, out)
        assert (got_mix[i] == out % np.uint64(P)).all(), i
    # complete addition on the special cases: P + P, P + (-P), P + O, O + O
    ident = np.array([0] * 6 + [1] + [0] * 5 + [0] * 6, np.uint64)
    neg = p.copy()
    for i in range(n):
        neg[i, 6:12] = ora.fp6_unop("ora_fp6_neg", p[i, 6:12])
    cases_a = np.stack([p[0], p[1], p[2], ident])
    cases_b = np.stack([p[0], neg[1], ident, ident])
    got = host(eng.debug_op(33, dev(cases_a), dev(cases_b), 18))
    for i in range(4):
        L.ora_proj_add(np.ascontiguousarray(cases_a[i]), np.ascontiguousarray(cases_b[i]), out)
        assert (got[i] == out % np.uint64(P)).all(), i


def test_modified_jacobian_formulas(eng, ora):
    L = ora.lib()
    n = 24
    pa = _random_points(ora, 47, n, "affine")
    qa = _random_points(ora, 49, n, "affine")
    lam = synth.fp6_elements(51, n)
    mj = np.zeros((n, 24), np.uint64)
    for i in range(n):
        l2 = ora.fp6_unop("ora_fp6_square", lam[i])
        l3 = ora.fp6_binop("ora_fp6_mul", l2, lam[i])
        mj[i, 0:6] = ora.fp6_binop("ora_fp6_mul", pa[i, 0:6], l2)
        mj[i, 6:12] = ora.fp6_binop("ora_fp6_mul", pa[i, 6:12], l3)
        mj[i, 12:18] = lam[i]
        mj[i, 18:24] = ora.fp6_unop("ora_fp6_square", l2)
    got_d = host(eng.debug_op(35, dev(mj), None, 24))
    got_a = host(eng.debug_op(36, dev(mj), dev(qa), 24))
    out = np.zeros(24, np.uint64)
    for i in range(n):
        L.ora_mjac_double_unchecked(np.ascontiguousarray(mj[i]), out)
        assert (got_d[i] == out % np.uint64(P)).all(), i
        L.ora_mjac_add_mixed_unchecked(np.ascontiguousarray(mj[i]), np.ascontiguousarray(qa[i]), 0, out)
        assert (got_a[i] == out % np.uint64(P)).all(), i


# ------------------------------------------------------------ Fp6 batches ----

def test_fp6_reference_kats_on_gpu(eng, kats):
    k = kats["fp6_mul"]
    assert list(eng.fp6_mul_batch(np.array([k["a"]], np.uint64), np.array([k["b"]], np.uint64))[0]) == k["out"]
    k = kats["fp6_square"]
    assert list(eng.fp6_square_batch(np.array([k["a"]], np.uint64))[0]) == k["out"]
    k = kats["fp6_invert"]
    out, some = eng.fp6_invert_batch(np.array([k["a"], [1, 0, 0, 0, 0, 0], [0] * 6, [P, 0, P, 0, 0, P]], np.uint64))
    assert list(out[0]) == k["out"] and list(out[1]) == [1, 0, 0, 0, 0, 0]
    assert list(some) == [1, 1, 0, 0] and not out[2].any() and not out[3].any()


def test_fp6_golden_model_vectors_on_gpu(eng, model_vectors):
    rows = model_vectors["fp6"]
    a = np.array([r["a"] for r in rows], np.uint64)
    b = np.array([r["b"] for r in rows], np.uint64)
    assert (eng.fp6_mul_batch(a, b) == np.array([r["mul"] for r in rows], np.uint64)).all()
    assert (eng.fp6_square_batch(a) == np.array([r["square"] for r in rows], np.uint64)).all()
    inv, some = eng.fp6_invert_batch(a)
    assert (inv == np.array([r["invert"] for r in rows], np.uint64)).all() and some.all()


@pytest.mark.parametrize("n", [1, 63, 256, 257, 4099])
def test_fp6_batches_vs_oracle_ragged(eng, ora, n):
    a, b = edge_fp6(n, 61), edge_fp6(n, 62)[::-1].copy()
    A, B = dev(a), dev(b)
    assert (host(eng.fp6_mul_batch(A, B)) == ora.fp6_mul_batch(a, b)).all()
    assert (host(eng.fp6_square_batch(A)) == ora.fp6_square_batch(a)).all()
    inv, some = eng.fp6_invert_batch(A)
    winv, wsome = ora.fp6_invert_batch(a)
    assert (host(inv) == winv).all() and (host(some) == wsome).all()
    # host-pointer entry points agree with the device ones
    assert (eng.fp6_mul_batch(a, b) == host(eng.fp6_mul_batch(A, B))).all()


def test_fp6_empty_and_errors(eng):
    from cheetah_amd import CheetahError
    z = np.zeros((0, 6), np.uint64)
    assert eng.fp6_mul_batch(z, z).shape == (0, 6)
    with pytest.raises(ValueError):
        eng.fp6_mul_batch(np.zeros((3, 5), np.uint64), np.zeros((3, 5), np.uint64))
    with pytest.raises(CheetahError):
        eng._check(eng._lib.cheetah_fp6_mul_batch(eng._h, None, None, None, 4))


def test_fp6_mul_full_size_config1(eng, ora):
    """BASELINE config 1: 2^20 seeded pairs, compared in full with the oracle."""
    n = 1 << 20
    A = eng.synth_fp6(1, n, 0)
    B = eng.synth_fp6(1, n, n)
    a, b = synth.fp6_elements(1, n, 0), synth.fp6_elements(1, n, n)
    assert (host(A) == a).all() and (host(B) == b).all()  # device generator == host generator
    got = host(eng.fp6_mul_batch(A, B))
    assert (got == ora.fp6_mul_batch(a, b)).all()
    # size-independent property: (a*b)*a^-1 == b wherever a != 0
    ainv, some = eng.fp6_invert_batch(A)
    back = host(eng.fp6_mul_batch(eng.fp6_mul_batch(A, B), ainv))
    assert host(some).all() and (back == b).all()


# ------------------------------------------------------------------ tables ----

def test_tables_vs_oracle(eng, ora, kats):
    bpt = eng.read_table(0)
    assert (bpt == ora.table("ora_basepoint_table", 256)).all()
    sh = eng.read_table(1)
    ms = ora.table("ora_minus_shift_point_array", 257)
    assert list(sh[0]) == kats["shift_point"]["x"] + kats["shift_point"]["y"]
    assert (sh[1] == ms[4]).all() and (sh[2] == ms[256]).all()
    assert list(bpt[0]) == kats["generator"]["x"] + kats["generator"]["y"]
    assert list(bpt[1]) == kats["two_g_affine"]["x"] + kats["two_g_affine"]["y"]  # 2G KAT, src/curve/projective.rs:1157-1178
    bpl = eng.read_table(2)  # BASEPOINT_LOOKUP = [G .. 128 G], src/constants.rs:26-27
    assert (bpl == ora.table("ora_basepoint_lookup", 128)).all()
    assert (bpl[:8] == bpt[:8]).all()


# --------------------------------------------------------- batch_normalize ----

@pytest.mark.parametrize("n", [1, 7, 8, 9, 300, 2049])
def test_batch_normalize_vs_oracle(eng, ora, n):
    aff = _random_points(ora, 71, min(n, 64), "affine")
    aff = np.tile(aff, (n // len(aff) + 1, 1))[:n]
    lam = synth.fp6_elements(72, n)
    jac = np.zeros((n, 18), np.uint64)
    prj = np.zeros((n, 18), np.uint64)
    for i in range(n):
        l2 = ora.fp6_unop("ora_fp6_square", lam[i])
        l3 = ora.fp6_binop("ora_fp6_mul", l2, lam[i])
        jac[i, 0:6] = ora.fp6_binop("ora_fp6_mul", aff[i, 0:6], l2)
        jac[i, 6:12] = ora.fp6_binop("ora_fp6_mul", aff[i, 6:12], l3)
        prj[i, 0:6] = ora.fp6_binop("ora_fp6_mul", aff[i, 0:6], lam[i])
        prj[i, 6:12] = ora.fp6_binop("ora_fp6_mul", aff[i, 6:12], lam[i])
        jac[i, 12:18] = lam[i]
        prj[i, 12:18] = lam[i]
    # identities sprinkled in (the case the reference's own test never takes),
    # including runs of them and non-canonical zeros
    rng = np.random.default_rng(n)
    idx = set(int(v) for v in rng.integers(0, n, max(1, n // 5)))
    if n >= 9:
        idx |= {0, 1, 2, n - 1}
    for i in idx:
        jac[i, 12:18] = 0
        prj[i, 12:18] = [P, 0, 0, P, 0, 0]
    from cheetah_amd import JACOBIAN, PROJECTIVE
    wxy, winf = ora.batch_normalize(jac, True)
    xy, inf = eng.batch_normalize(dev(jac), JACOBIAN)
    assert (host(xy) == wxy).all() and (host(inf) == winf).all()
    assert winf.sum() == len(idx)
    wxy, winf = ora.batch_normalize(prj, False)
    xy, inf = eng.batch_normalize(dev(prj), PROJECTIVE)
    assert (host(xy) == wxy).all() and (host(inf) == winf).all()
    xy, inf = eng.batch_normalize(prj, PROJECTIVE)  # host-pointer entry point
    assert (xy == wxy).all() and (inf == winf).all()


def test_batch_normalize_golden(eng, model_vectors):
    from cheetah_amd import JACOBIAN, PROJECTIVE
    rows = model_vectors["normalize"]
    want = np.array([r["affine"]["xy"] for r in rows], np.uint64)
    xy, inf = eng.batch_normalize(np.array([r["jacobian"] for r in rows], np.uint64), JACOBIAN)
    assert (xy == want).all() and not inf.any()
    xy, inf = eng.batch_normalize(np.array([r["projective"] for r in rows], np.uint64), PROJECTIVE)
    assert (xy == want).all() and not inf.any()
    assert eng.batch_normalize(np.zeros((0, 18), np.uint64), JACOBIAN)[0].shape == (0, 12)


# -------------------------------------------------------------- fixed base ----

def test_mul_by_generator_golden_and_oracle(eng, ora, model_vectors):
    from cheetah_amd import AFFINE, PROJECTIVE, CheetahError
    rows = [r for r in model_vectors["mul_by_generator"] if r["scalar"][31] >> 4 <= 7]
    sc = np.array([r["scalar"] for r in rows], np.uint8)
    xy, inf = eng.mul_by_generator_batch(sc, AFFINE)
    assert [list(v) for v in xy] == [r["out"]["xy"] for r in rows]
    assert list(inf) == [r["out"]["inf"] for r in rows]
    # the reference's own projective representative (L2)
    pr, pinf = eng.mul_by_generator_batch(sc, PROJECTIVE)
    wpr, wpinf = ora.mul_by_generator_batch(sc, 1)
    assert (pr == wpr).all() and (pinf == wpinf).all()
    # out-of-range scalar (top nibble > 7): rejected by the host entry point
    bad = sc.copy()
    bad[0, 31] = 0x80
    with pytest.raises(CheetahError) as ei:
        eng.mul_by_generator_batch(bad, AFFINE)
    assert ei.value.code == -4


@pytest.mark.parametrize("n", [1, 255, 1000])
def test_mul_by_generator_seeded_vs_oracle(eng, ora, n):
    from cheetah_amd import AFFINE, PROJECTIVE
    sc = synth.scalars(2, n)
    S = dev(sc)
    xy, inf = eng.mul_by_generator_batch(S, AFFINE)
    wxy, winf = ora.mul_by_generator_batch(sc, 0)
    assert (host(xy) == wxy).all() and (host(inf) == winf).all()
    pr, _ = eng.mul_by_generator_batch(S, PROJECTIVE)
    assert (host(pr) == ora.mul_by_generator_batch(sc, 1)[0]).all()
    assert (host(eng.synth_scalars(2, n)) == sc).all()


def test_mul_by_generator_wide_comb_edges(eng, ora):
    """The affine-result fixed-base kernel (signed radix-64 comb, 43 windows, no doublings) against the
    oracle's BASEPOINT_TABLE.multiply on scalars that stress its recoding: digits at the re-centring
    boundary (31 / 32 / 33 / 63 in every window), carry chains through all windows, 0, 1, q - 1, q
    (identity) and the largest scalar the reference's comb covers (2^255 - 1)."""
    from cheetah_amd import AFFINE
    ks = [0, 1, 2, 31, 32, 33, 63, 64, 65, Q - 1, Q, Q + 1, (1 << 255) - 1, (1 << 255) - 64, (1 << 252) - 1, 1 << 252]
    for j in range(43):
        for d in (31, 32, 33, 63):
            k = d << (6 * j)
            if k < (1 << 255):
                ks.append(k)
    ks += [int("ff" * 31, 16), int("df" * 31 + "20", 16) & ((1 << 255) - 1), sum(32 << (6 * j) for j in range(42)),
           sum(63 << (6 * j) for j in range(42)), sum(31 << (6 * j) for j in range(42))]
    assert all(k < (1 << 255) for k in ks)  # the range BasePointTable::multiply covers
    sc = np.array([np.frombuffer(int(k).to_bytes(32, "little"), np.uint8) for k in ks], np.uint8)
    xy, inf = eng.mul_by_generator_batch(dev(sc), AFFINE)
    wxy, winf = ora.mul_by_generator_batch(sc, 0)
    bad = [hex(ks[i]) for i in range(len(ks)) if not ((host(xy)[i] == wxy[i]).all() and host(inf)[i] == winf[i])]
    assert not bad, bad
    assert host(inf)[0] == 1 and host(inf)[ks.index(Q)] == 1  # 0 G = q G = identity


def test_basepoint_lookup_jacobian(eng, ora):
    """`&BASEPOINT_LOOKUP * &Scalar` (src/constants.rs:102-135): the reference's Jacobian
    (X, Y, Z), edge scalars included (digit 31 wraps to -128 for a top byte of 0x7f + carry,
    as the reference's `as i8` does), and BASEPOINT_LOOKUP * c == BASEPOINT_TABLE * c
    (src/curve/jacobian.rs:1612-1626)."""
    from cheetah_amd import AFFINE, JACOBIAN
    n = 300
    sc = synth.scalars(6, n)
    edge = [0, 1, 2, 127, 128, 129, 255, 256, Q - 1, Q, (1 << 255) - 1, (0x7f << 248) | (0x80 << 240), (1 << 256) - 1]
    for i, k in enumerate(edge):
        sc[i] = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8)
    S = dev(sc)
    jac, jinf = eng.mul_by_generator_batch(S, JACOBIAN)
    wjac, wjinf = ora.mul_by_generator_batch(sc, 2)
    assert (host(jac) == wjac).all() and (host(jinf) == wjinf).all()
    from cheetah_amd import JACOBIAN as J
    xy, inf = eng.batch_normalize(jac, J)
    can = np.array([int.from_bytes(bytes(s), "little") < Q for s in sc])
    txy, tinf = eng.mul_by_generator_batch(dev(sc[can]), AFFINE)
    assert (host(xy)[can] == host(txy)).all() and (host(inf)[can] == host(tinf)).all()
    hj, hi = eng.mul_by_generator_batch(sc, JACOBIAN)  # host-pointer entry point takes any 32 bytes on this path
    assert (hj == wjac).all() and (hi == wjinf).all()


# ----------------------------------------------------------- variable base ----

def test_multiply_golden(eng, model_vectors):
    from cheetah_amd import AFFINE
    rows = model_vectors["multiply"]
    pts = np.array([r["point"]["xy"] for r in rows], np.uint64)
    sc = np.array([r["scalar"] for r in rows], np.uint8)
    want_xy = [r["out"]["xy"] for r in rows]
    want_inf = [r["out"]["inf"] for r in rows]
    xy, inf = eng.multiply_batch(pts, sc, AFFINE, AFFINE)
    assert [list(v) for v in xy] == want_xy and list(inf) == want_inf
    xy, inf = eng.multiply_vartime_batch(pts, sc, AFFINE, AFFINE)
    for i, r in enumerate(rows):
        if int.from_bytes(bytes(sc[i]), "little") < 2**255:  # a set bit 255 is outside the 256-digit NAF
            assert list(xy[i]) == want_xy[i] and inf[i] == want_inf[i], i


@pytest.mark.parametrize("n", [1, 64, 300])
def test_multiply_vs_oracle_all_forms(eng, ora, n):
    """ct and vartime ladders; affine and projective inputs; affine output (L1) and
    the reference's projective representative (L2); multiply == multiply_vartime
    (src/curve/projective.rs:1339)."""
    from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE, CheetahError
    sc = synth.scalars(3, n)
    aff = _random_points(ora, 81, n, "affine")
    prj = _random_points(ora, 81, n, "proj")
    # edge scalars and an identity point
    edge = [0, 1, 2, 15, 16, Q - 1, Q, (1 << 255) - 1]
    for i, k in enumerate(edge[: n]):
        sc[i] = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8)
    if n >= 64:
        aff[20] = [0] * 6 + [1] + [0] * 5       # affine identity encoding
        prj[20, 12:18] = 0                      # projective identity
        sc[21] = 255                            # non-canonical 2^256 - 1: needs LookupTable<16>'s entry 16
    S = dev(sc)
    jac = _random_points(ora, 81, n, "jac")
    if n >= 64:
        jac[20, 12:18] = 0                      # Jacobian identity
    if n >= 300:                                # the same in a wave without a wide scalar (multiply_vartime's AFFINE results take the
        aff[100] = [0] * 6 + [1] + [0] * 5      # fixed-window path there): (X, Y, 0) with X != 0 is an identity whose LookupTable
        prj[100, 12:18] = 0                     # entries are arbitrary values and whose NafLookupTable entries are identities
        jac[100, 12:18] = 0
        prj[101] = [0] * 6 + [1] + [0] * 5 + [0] * 6
    for vartime in (False, True):
        f = eng.multiply_vartime_batch if vartime else eng.multiply_batch
        for in_form, pts in ((AFFINE, aff), (PROJECTIVE, prj)):
            Pd = dev(pts)
            wxy, winf = ora.multiply_batch(pts, in_form, sc, 0, vartime=vartime)
            xy, inf = f(Pd, S, in_form, AFFINE)
            assert (host(xy) == wxy).all() and (host(inf) == winf).all(), (vartime, in_form)
            wpr, wpinf = ora.multiply_batch(pts, in_form, sc, 1, vartime=vartime)
            pr, pinf = f(Pd, S, in_form, PROJECTIVE)
            assert (host(pr) == wpr).all() and (host(pinf) == wpinf).all(), (vartime, in_form)
        # JacobianPoint::multiply* (src/curve/jacobian.rs:599-649): the reference's own Jacobian (X, Y, Z)
        for in_form, pts in ((AFFINE, aff), (JACOBIAN, jac)):
            Pd = dev(pts)
            wj, wjinf = ora.multiply_batch(pts, in_form, sc, 2, vartime=vartime)
            j, jinf = f(Pd, S, in_form, JACOBIAN)
            assert (host(j) == wj).all() and (host(jinf) == wjinf).all(), (vartime, in_form)
        wxy, winf = ora.multiply_batch(jac, JACOBIAN, sc, 0, vartime=vartime)
        xy, inf = f(dev(jac), S, JACOBIAN, AFFINE)
        assert (host(xy) == wxy).all() and (host(inf) == winf).all(), vartime
        with pytest.raises(CheetahError):
            f(dev(prj), S, PROJECTIVE, JACOBIAN)
    # cross-path: ct == vartime on canonical scalars
    can = np.array([int.from_bytes(bytes(s), "little") < 2**255 for s in sc])
    a1, i1 = eng.multiply_batch(dev(aff), S, AFFINE, AFFINE)
    a2, i2 = eng.multiply_vartime_batch(dev(aff), S, AFFINE, AFFINE)
    assert (host(a1)[can] == host(a2)[can]).all() and (host(i1)[can] == host(i2)[can]).all()
    # host-pointer entry point
    xy, inf = eng.multiply_batch(aff, sc, AFFINE, AFFINE)
    assert (xy == host(a1)).all() and (inf == host(i1)).all()


def test_multiply_vartime_mixed_wide_scalars(eng, ora):
    """multiply_vartime with AFFINE results splits a batch between two kernels: waves whose scalars are all
    below 2^255 take the fixed-window ladder, a wave holding any scalar with bit 255 set takes the reference's
    NAF ladder (whose recoding drops the last carry there).  Random 256-bit scalars, about one in eight with
    bit 255 set, so that both kinds of wave occur and mix inside workgroups; against the oracle."""
    from cheetah_amd import AFFINE, PROJECTIVE
    n = 700
    rng = np.random.default_rng(2024)
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x7F
    wide = rng.random(n) < 0.125
    wide[64:256] = False          # waves 1-3 of workgroup 0: all narrow
    wide[256:320] = True          # wave 0 of workgroup 1: all wide
    sc[wide, 31] |= 0x80
    pts = _random_points(ora, 91, n, "affine")
    xy, inf = eng.multiply_vartime_batch(dev(pts), dev(sc), AFFINE, AFFINE)
    wxy, winf = ora.multiply_batch(pts, 0, sc, 0, vartime=True)
    bad = [i for i in range(n) if not ((host(xy)[i] == wxy[i]).all() and host(inf)[i] == winf[i])]
    assert not bad, (bad[:10], [bool(wide[i]) for i in bad[:10]])
    pr, pinf = eng.multiply_vartime_batch(dev(pts), dev(sc), AFFINE, PROJECTIVE)
    wpr, wpinf = ora.multiply_batch(pts, 0, sc, 1, vartime=True)
    assert (host(pr) == wpr).all() and (host(pinf) == wpinf).all()


def test_reserved_cus_same_results(eng, ora):
    """cheetah_ctx_set_reserved_cus only sizes the persistent launches smaller (room for an RCCL kernel on another
    stream): same results, fewer units per launch; out-of-range values are rejected."""
    from cheetah_amd import AFFINE, CheetahError
    n = 3000
    sc = synth.scalars(12, n)
    pts, _ = eng.mul_by_generator_batch(dev(sc), AFFINE)
    k = dev(synth.scalars(13, n))
    full = eng.multiply_units_per_launch(1 << 26)
    a0, i0 = eng.multiply_batch(pts, k, AFFINE, AFFINE)
    try:
        eng.set_reserved_cus(8)
        assert eng.multiply_units_per_launch(1 << 26) < full
        a1, i1 = eng.multiply_batch(pts, k, AFFINE, AFFINE)
        g1, _ = eng.mul_by_generator_batch(dev(sc), AFFINE)
        with pytest.raises(CheetahError):
            eng.set_reserved_cus(100000)
    finally:
        eng.set_reserved_cus(0)
    assert (host(a0) == host(a1)).all() and (host(i0) == host(i1)).all() and (host(g1) == host(pts)).all()
    assert eng.multiply_units_per_launch(1 << 26) == full


def test_launch_rounds_same_results(eng):
    """cheetah_ctx_set_launch_rounds only changes how many rounds of the resident grid one launch covers: same results
    over a batch that spans several launches at one round each; out-of-range values are rejected."""
    from cheetah_amd import AFFINE, CheetahError
    n = (1 << 19) + 777
    sc = eng.synth_scalars(31, n)
    pts, _ = eng.mul_by_generator_batch(sc, AFFINE)
    k = eng.synth_scalars(32, n)
    two = eng.multiply_units_per_launch(1 << 26)
    a0, i0 = eng.multiply_batch(pts, k, AFFINE, AFFINE)
    try:
        eng.set_launch_rounds(1)
        assert eng.multiply_units_per_launch(1 << 26) * 2 == two
        a1, i1 = eng.multiply_batch(pts, k, AFFINE, AFFINE)
        eng.set_launch_rounds(16)
        assert eng.multiply_units_per_launch(1 << 26) == two * 8
        a2, i2 = eng.multiply_batch(pts, k, AFFINE, AFFINE)
        g2, _ = eng.mul_by_generator_batch(sc, AFFINE)
        for bad in (0, 257, -1):
            with pytest.raises(CheetahError):
                eng.set_launch_rounds(bad)
    finally:
        eng.set_launch_rounds(2)
    assert (host(a0) == host(a1)).all() and (host(a0) == host(a2)).all() and (host(i0) == host(i1)).all() and (host(i0) == host(i2)).all()
    assert (host(g2) == host(pts)).all()
    assert eng.multiply_units_per_launch(1 << 26) == two


def test_torsion_and_cofactor_kats(eng, kats):
    """q * G = O through the full vartime ladder (is_torsion_free, src/curve/projective.rs:885-893, :1462);
    the reference's non-subgroup point is not killed by q but its cofactor multiple is (:1427-1433)."""
    from cheetah_amd import AFFINE
    g = kats["generator"]["x"] + kats["generator"]["y"]
    ns = kats["non_subgroup_point"]["x"] + kats["non_subgroup_point"]["y"]
    qb, hb = kats["fq_modulus_bytes"], kats["cofactor_bytes"]
    pts = np.array([g, ns, ns], np.uint64)
    sc = np.array([qb, qb, hb], np.uint8)
    for f in (eng.multiply_vartime_batch, eng.multiply_batch):
        xy, inf = f(pts, sc, AFFINE, AFFINE)
        assert list(inf) == [1, 0, 0]
        assert list(xy[0]) == [0] * 6 + [1] + [0] * 5
        xy2, inf2 = f(np.ascontiguousarray(xy[2:3]), np.array([qb], np.uint8), AFFINE, AFFINE)
        assert list(inf2) == [1]


def test_g_a_b_equals_g_ab_on_gpu(eng):
    # (g*a)*b == g*(a*b), src/curve/projective.rs:1338, and == BASEPOINT_TABLE path (:1341)
    from cheetah_amd import AFFINE
    n = 200
    a, b = synth.scalars(91, n), synth.scalars(92, n)
    ab = np.array([list(((int.from_bytes(bytes(x), "little") * int.from_bytes(bytes(y), "little")) % Q).to_bytes(32, "little"))
                   for x, y in zip(a, b)], np.uint8)
    ga, _ = eng.mul_by_generator_batch(dev(a), AFFINE)
    gab, i1 = eng.multiply_batch(ga, dev(b), AFFINE, AFFINE)
    want, i2 = eng.mul_by_generator_batch(dev(ab), AFFINE)
    assert (host(gab) == host(want)).all() and (host(i1) == host(i2)).all()



def test_fp_op_batch_vs_python_ints(eng, ora):
    """cheetah_fp_op_batch: the base field's operations (src/fp.rs:125-196, :335-359) against Python integers mod p and the
    oracle, edge values and every pair of them included; any u64 is a legal input, outputs are canonical."""
    rng = np.random.default_rng(9)
    edge = [0, 1, 2, P - 2, P - 1, P, P + 1, 2**32 - 1, 2**32, 2**32 + 1, 2**63, 2**64 - 2**32, 2**64 - 1]
    pairs = [(x, y) for x in edge for y in edge]
    a = np.array([x for x, _ in pairs] + list(rng.integers(0, 2**64, 3000, dtype=np.uint64)), np.uint64)
    b = np.array([y for _, y in pairs] + list(rng.integers(0, 2**64, 3000, dtype=np.uint64)), np.uint64)
    ai, bi = [int(v) for v in a], [int(v) for v in b]
    want = {
        "add": [(x + y) % P for x, y in zip(ai, bi)], "sub": [(x - y) % P for x, y in zip(ai, bi)], "neg": [(-x) % P for x in ai],
        "double": [2 * x % P for x in ai], "triple": [3 * x % P for x in ai], "mul": [x * y % P for x, y in zip(ai, bi)],
        "square": [x * x % P for x in ai], "canonical": [x % P for x in ai],
    }
    A, B = dev(a), dev(b)
    for op, w in want.items():
        binary = op in ("add", "sub", "mul")
        w = np.array(w, np.uint64)
        assert (host(eng.fp_op_batch(op, A, B if binary else None)) == w).all(), op
        assert (eng.fp_op_batch(op, a, b if binary else None) == w).all(), (op, "host")
    for k in (0, 1, 7, 2**31, 2**32 - 1):
        w = np.array([x * k % P for x in ai], np.uint64)
        assert (host(eng.fp_op_batch("mul_by_u32", A, scalar=k)) == w).all(), k
        assert int(ora.lib().ora_fp_mul_by_u32(ai[20], k)) % P == int(w[20])
    inv, some = eng.fp_op_batch("invert", A)
    inv, some = host(inv), host(some)
    w = np.array([pow(x, P - 2, P) for x in ai], np.uint64)
    assert (inv == w).all() and (some == np.array([x % P != 0 for x in ai], np.uint8)).all()
    inv2, some2 = eng.fp_op_batch("invert", a)
    assert (inv2 == w).all() and (some2 == some).all()


def test_fp6_op_batch_vs_oracle(eng, ora):
    """cheetah_fp6_op_batch: the coefficient-wise operations, the Frobenius maps and mul_by_3b (src/fp6.rs:373-382, :428-464,
    :968-1029, :1088-1123, src/curve/mod.rs:54-106) on seeded elements plus edge limbs (0, 1, p - 1, p, 2^64 - 1), canonical
    output; device and host arrays."""
    n = 300
    a, b = synth.fp6_elements(61, n).reshape(n, 6).copy(), synth.fp6_elements(62, n).reshape(n, 6).copy()
    edge = np.array([0, 1, P - 1, P, 2**64 - 1, 2**32, 2**32 - 1], np.uint64)
    a[:7, :] = edge[:, None]
    b[:7, :] = edge[::-1][:, None]
    a[7] = [2**64 - 1, P, 0, 1, P - 1, 2**63]
    L = ora.lib()
    def want(f, *extra):
        out = np.zeros((n, 6), np.uint64)
        for i in range(n):
            o = np.zeros(6, np.uint64)
            f(i, o)
            out[i] = o % np.uint64(P)
        return out
    cases = [
        ("add", True, 0, lambda i, o: L.ora_fp6_add(a[i].copy(), b[i].copy(), o)),
        ("sub", True, 0, lambda i, o: L.ora_fp6_sub(a[i].copy(), b[i].copy(), o)),
        ("neg", False, 0, lambda i, o: L.ora_fp6_neg(a[i].copy(), o)),
        ("double", False, 0, lambda i, o: L.ora_fp6_double(a[i].copy(), o)),
        ("triple", False, 0, lambda i, o: L.ora_fp6_triple(a[i].copy(), o)),
        ("mul_by_u32", False, 0xFFFFFFFF, lambda i, o: L.ora_fp6_mul_by_u32(a[i].copy(), 0xFFFFFFFF, o)),
        ("mul_by_u32", False, 7, lambda i, o: L.ora_fp6_mul_by_u32(a[i].copy(), 7, o)),
        ("mul_by_fp", False, 2**64 - 1, lambda i, o: L.ora_fp6_mul_by_fp(a[i].copy(), 2**64 - 1, o)),
        ("mul_by_fp", False, 0x123456789ABCDEF1, lambda i, o: L.ora_fp6_mul_by_fp(a[i].copy(), 0x123456789ABCDEF1, o)),
        ("canonical", False, 0, lambda i, o: L.ora_fp6_canonical(a[i].copy(), o)),
        ("frobenius", False, 0, lambda i, o: L.ora_fp6_frobenius(a[i].copy(), o)),
        ("frobenius_double", False, 0, lambda i, o: L.ora_fp6_frobenius_double(a[i].copy(), o)),
        ("frobenius_triple", False, 0, lambda i, o: L.ora_fp6_frobenius_triple(a[i].copy(), o)),
        ("mul_by_3b", False, 0, lambda i, o: L.ora_mul_by_3b(a[i].copy(), o)),
    ]
    A, B = dev(a), dev(b)
    for op, binary, scalar, f in cases:
        w = want(f)
        got = eng.fp6_op_batch(op, A, B if binary else None, scalar)
        assert (host(got) == w).all(), (op, scalar)
        got = eng.fp6_op_batch(op, a, b if binary else None, scalar)
        assert (got == w).all(), (op, scalar, "host")
    # x^(p^6) = x: three Frobenius maps twice over
    x = A
    for _ in range(2):
        x = eng.fp6_op_batch("frobenius_triple", x)
    assert (host(x) == a % np.uint64(P)).all()


# ------------------------------------------ multi-scalar ladders and checks ----

@pytest.mark.parametrize("m", [1, 2, 3])
def test_multiply_many_vs_oracle(eng, ora, m):
    """multiply_many / multiply_many_vartime (m = 2: multiply_double*), all forms, the reference's
    own (X, Y, Z) for projective / Jacobian outputs; and the relation the reference tests:
    sum_j k_j P_j computed jointly == the sum of separate products (src/curve/projective.rs:1354-1397)."""
    from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE
    n = 70
    sc = synth.scalars(7, n * m)
    edge = [0, 1, Q - 1, 16, (1 << 255) - 1]
    for i, k in enumerate(edge):
        sc[i] = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8)
    # bit 255 set (no canonical scalar): the radix-16 and NAF ladders differ there; the vartime entry points'
    # affine fast path must hand those waves to the NAF kernel
    sc[n * m - 3] = np.frombuffer(int((1 << 255) + 77).to_bytes(32, "little"), np.uint8)
    sc[n * m - 1] = np.frombuffer(int((1 << 256) - 1).to_bytes(32, "little"), np.uint8)
    aff = _random_points(ora, 83, n * m, "affine")
    prj = _random_points(ora, 83, n * m, "proj")
    jac = _random_points(ora, 83, n * m, "jac")
    aff[9] = [0] * 6 + [1] + [0] * 5
    prj[9, 12:18] = 0
    jac[9, 12:18] = 0
    S = dev(sc)
    for vartime in (False, True):
        f = eng.multiply_many_vartime_batch if vartime else eng.multiply_many_batch
        for in_form, pts, out_forms in ((AFFINE, aff, (AFFINE, PROJECTIVE, JACOBIAN)), (PROJECTIVE, prj, (AFFINE, PROJECTIVE)),
                                        (JACOBIAN, jac, (AFFINE, JACOBIAN))):
            for out_form in out_forms:
                want, winf = ora.multiply_many_batch(pts, in_form, sc, m, out_form, vartime=vartime)
                got, inf = f(dev(pts), S, m, in_form, out_form)
                assert (host(got) == want).all() and (host(inf) == winf).all(), (vartime, in_form, out_form)
    # host-pointer entry point
    got, inf = eng.multiply_many_batch(aff, sc, m, AFFINE, AFFINE)
    want, winf = ora.multiply_many_batch(aff, 0, sc, m, 0)
    assert (got == want).all() and (inf == winf).all()
    if m == 2:  # a*P + b*P == (a + b)*P
        P2 = np.repeat(aff[:n], 2, axis=0)
        got, _ = eng.multiply_many_batch(P2, sc, 2, AFFINE, AFFINE)
        ab = np.array([list(((int.from_bytes(bytes(sc[2 * i]), "little") + int.from_bytes(bytes(sc[2 * i + 1]), "little")) % Q).to_bytes(32, "little"))
                       for i in range(n)], np.uint8)
        red = np.array([int.from_bytes(bytes(sc[2 * i]), "little") < Q and int.from_bytes(bytes(sc[2 * i + 1]), "little") < Q for i in range(n)])
        single, _ = eng.multiply_batch(aff[:n], ab, AFFINE, AFFINE)
        assert (got[red] == single[red]).all()


def test_multiply_double_with_basepoint_vartime(eng, ora):
    """a*P + b*G through ODD_MULTIPLES_BASEPOINT (src/curve/projective.rs:756-809); the table itself
    against the oracle's NafLookupTable<64> of G."""
    from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE
    assert (eng.read_table(3) == ora.table("ora_odd_multiples_basepoint", 64)).all()
    n = 130
    sc = synth.scalars(8, 2 * n)
    for i, k in enumerate([0, 0, 1, 0, 0, 1, Q - 1, Q - 1]):
        sc[i] = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8)
    aff = _random_points(ora, 85, n, "affine")
    prj = _random_points(ora, 85, n, "proj")
    jac = _random_points(ora, 85, n, "jac")
    for in_form, pts, out_forms in ((AFFINE, aff, (AFFINE, JACOBIAN)), (PROJECTIVE, prj, (AFFINE, PROJECTIVE)), (JACOBIAN, jac, (JACOBIAN,))):
        for out_form in out_forms:
            want, winf = ora.multiply_many_batch(pts, in_form, sc, 1, out_form, vartime=True, with_basepoint=True)
            got, inf = eng.multiply_double_with_basepoint_vartime_batch(dev(pts), dev(sc), in_form, out_form)
            assert (host(got) == want).all() and (host(inf) == winf).all(), (in_form, out_form)
    got, inf = eng.multiply_double_with_basepoint_vartime_batch(aff, sc, AFFINE, AFFINE)
    want, winf = ora.multiply_many_batch(aff, 0, sc, 1, 0, vartime=True, with_basepoint=True)
    assert (got == want).all() and (inf == winf).all()


def test_multiply_double_with_basepoint_composed_path(eng, ora, kats):
    """The AFFINE results of multiply_double_with_basepoint_vartime come from a composed schedule (a P by the
    fixed-window ladder, b G by the widened comb, one complete addition, k_affine_sum): against the oracle's restatement
    of the reference's Straus ladder on the cases the final addition selects on (a P = b G, a P = -b G, either or both the
    identity, P the identity), with scalars that have bit 255 set in either position (those waves are redone by the NAF
    kernel), for the three input forms; and against the reference schedule on the GPU at 2^18."""
    import torch
    from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE
    n = 600
    le = lambda k: np.frombuffer(int(k).to_bytes(32, "little"), np.uint8)
    sc = synth.scalars(18, 2 * n).reshape(n, 2, 32)
    aff = _random_points(ora, 95, n, "affine")
    g = np.array(kats["generator"]["x"] + kats["generator"]["y"], np.uint64)
    neg_g = g.copy()
    neg_g[6:] = (np.uint64(P) - g[6:]) % np.uint64(P)
    ident = np.array([0] * 6 + [1] + [0] * 5, np.uint64)
    cases = [(g, 5, 5), (g, 5, Q - 5), (neg_g, 7, 7), (neg_g, 9, Q - 9), (g, 0, 0), (g, 0, 11), (g, 11, 0), (ident, 3, 4),
             (ident, 3, 0), (g, Q - 1, 1), (g, 1, Q - 1), (g, (1 << 255) - 1, 2)]
    for i, (pt, a, b) in enumerate(cases):
        j = 64 + 3 * i                       # in waves that hold no wide scalar
        aff[j] = pt
        sc[j, 0], sc[j, 1] = le(a), le(b)
    wide = [((1 << 255) + 12345, 7), (7, (1 << 255) + 99), ((1 << 256) - 1, (1 << 256) - 1), (1 << 255, 1 << 255)]
    for i, (a, b) in enumerate(wide):        # waves 4.. of the batch: redone by the NAF kernel, neighbours included
        j = 256 + 64 * i + 5
        sc[j, 0], sc[j, 1] = le(a), le(b)
    sc = np.ascontiguousarray(sc.reshape(2 * n, 32))
    prj = _to_form(ora, aff, "proj", 96)
    jac = _to_form(ora, aff, "jac", 96)
    for in_form, pts in ((AFFINE, aff), (PROJECTIVE, prj), (JACOBIAN, jac)):
        want, winf = ora.multiply_many_batch(pts, in_form, sc, 1, AFFINE, vartime=True, with_basepoint=True)
        got, inf = eng.multiply_double_with_basepoint_vartime_batch(dev(pts), dev(sc), in_form, AFFINE)
        assert (host(got) == want).all() and (host(inf) == winf).all(), in_form
        got, inf = eng.multiply_double_with_basepoint_vartime_batch(pts, sc, in_form, AFFINE)
        assert (got == want).all() and (inf == winf).all(), in_form
    # the selections did what they are for
    j = 64
    assert winf[j + 3] == 1 and winf[j + 6] == 1 and winf[j + 12] == 1     # 5G - 5G, -7G + 7G, 0 + 0
    # 2^18 pairs: composed == reference schedule
    N = 1 << 18
    k = eng.synth_scalars(19, 2 * N)
    pts, _ = eng.mul_by_generator_batch(eng.synth_scalars(20, N), AFFINE)
    a, ai = eng.multiply_double_with_basepoint_vartime_batch(pts, k, AFFINE, AFFINE)
    eng.set_reference_schedule(True)
    try:
        b, bi = eng.multiply_double_with_basepoint_vartime_batch(pts, k, AFFINE, AFFINE)
    finally:
        eng.set_reference_schedule(False)
    assert torch.equal(a.view(torch.int64), b.view(torch.int64)) and torch.equal(ai, bi)
    # multiply_many_vartime, m = 2 and 3: the indexed fixed-window ladder == the NAF ladder on canonical scalars
    for m in (2, 3):
        G = N // 4
        km = eng.synth_scalars(21 + m, G * m)
        pm = pts[: G * m]
        a, ai = eng.multiply_many_vartime_batch(pm, km, m, AFFINE, AFFINE)
        eng.set_reference_schedule(True)
        try:
            b, bi = eng.multiply_many_vartime_batch(pm, km, m, AFFINE, AFFINE)
        finally:
            eng.set_reference_schedule(False)
        assert torch.equal(a.view(torch.int64), b.view(torch.int64)) and torch.equal(ai, bi), m


def test_is_torsion_free_and_clear_cofactor(eng, ora, kats):
    """is_torsion_free / clear_cofactor (src/curve/projective.rs:872-893): G and seeded subgroup points are
    torsion free, the reference's non-subgroup point (:1407-1425) is not, its cofactor multiple is;
    clear_cofactor == multiply_vartime by the cofactor bytes, representative included."""
    from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE
    g = kats["generator"]["x"] + kats["generator"]["y"]
    ns = kats["non_subgroup_point"]["x"] + kats["non_subgroup_point"]["y"]
    sub = _random_points(ora, 87, 100, "affine")
    pts = np.concatenate([np.array([g, ns, [0] * 6 + [1] + [0] * 5], np.uint64), sub])
    flags = eng.is_torsion_free_batch(pts, AFFINE)
    assert list(flags[:3]) == [1, 0, 1] and flags[3:].all()
    assert (host(eng.is_torsion_free_batch(dev(pts), AFFINE)) == flags).all()
    prj = _random_points(ora, 87, 100, "proj")
    assert host(eng.is_torsion_free_batch(dev(prj), PROJECTIVE)).all()
    jac = _random_points(ora, 87, 100, "jac")
    assert host(eng.is_torsion_free_batch(dev(jac), JACOBIAN)).all()
    hb = np.tile(np.array(kats["cofactor_bytes"], np.uint8), (len(pts), 1))
    for in_form, p, out_form in ((AFFINE, pts, AFFINE), (AFFINE, pts, JACOBIAN), (PROJECTIVE, prj, PROJECTIVE), (JACOBIAN, jac, JACOBIAN)):
        want, winf = ora.multiply_batch(p, in_form, hb[: len(p)], out_form, vartime=True)
        got, inf = eng.clear_cofactor_batch(dev(p), in_form, out_form)
        assert (host(got) == want).all() and (host(inf) == winf).all(), (in_form, out_form)
    cleared, cinf = eng.clear_cofactor_batch(pts, AFFINE, AFFINE)
    assert eng.is_torsion_free_batch(cleared, AFFINE).all()



# ---------------------------------------------------------------- encodings ----

@pytest.mark.parametrize("n", [1, 255, 256, 1031])
def test_encodings_vs_oracle(eng, ora, kats, n):
    """to_uncompressed / to_compressed / from_uncompressed(_unchecked) (src/curve/encoding.rs), ragged sizes,
    identities, the reference's identity byte strings, and every rejection case."""
    xy, inf = ora.mul_by_generator_batch(synth.scalars(23, n), 0)
    xy, inf = xy.copy(), inf.copy()
    if n > 5:
        xy[3] = [0] * 6 + [1] + [0] * 5
        inf[3] = 1
        xy[4, 0] = np.uint64(int(xy[4, 0]) + P) if int(xy[4, 0]) + P < 2**64 else xy[4, 0]   # non-canonical limb in
    for compressed in (False, True):
        want = ora.encode_batch(xy, inf, compressed)
        f = eng.to_compressed_batch if compressed else eng.to_uncompressed_batch
        assert (host(f(dev(xy), dev(inf))) == want).all()
        assert (f(xy, inf) == want).all()
    if n > 5:
        assert list(eng.to_compressed_batch(xy, inf)[3]) == kats["identity_encodings"]["compressed"]
        assert list(eng.to_uncompressed_batch(xy, inf)[3]) == kats["identity_encodings"]["uncompressed"]
    enc = ora.encode_batch(xy, inf, False)
    if n > 5:
        ns = np.array([kats["non_subgroup_point"]["x"] + kats["non_subgroup_point"]["y"]], np.uint64)
        enc[0, 0:8] = 0xFF                                        # non-canonical limb: none
        enc[1, :] = 0                                             # (0, 0) without the flag: unchecked accepts, checked rejects
        enc[2, 96] = 0x80                                         # flag with non-zero coordinates: none
        enc[5] = ora.encode_batch(ns, np.zeros(1, np.uint8), False)[0]   # on the curve, outside the subgroup
    for checked in (False, True):
        wxy, winf, wsome = ora.decode_uncompressed_batch(enc, checked)
        gxy, ginf, gsome = eng.from_uncompressed_batch(dev(enc), checked)
        assert (host(gxy) == wxy).all() and (host(ginf) == winf).all() and (host(gsome) == wsome).all(), checked
        hxy, hinf, hsome = eng.from_uncompressed_batch(enc, checked)
        assert (hxy == wxy).all() and (hinf == winf).all() and (hsome == wsome).all()
    if n > 5:
        assert list(wsome[:6]) == [0, 0, 0, 1, 1, 0]


@pytest.mark.parametrize("n", [1, 255, 777])
def test_sqrt_decompression_from_x_vs_oracle(eng, ora, kats, n):
    """Fp6::sqrt (src/fp6.rs:922-964 over Fp3::sqrt src/fp3.rs:247-277), CompressedPoint::to_affine /
    AffinePoint::from_compressed (src/curve/encoding.rs:82-129, src/curve/affine.rs:303-309), AffinePoint::from_x
    (:326-343): squares, non-squares, edge limbs, the reference's rejection KATs, ragged sizes; bit-exact vs the oracle."""
    a = edge_fp6(n, 61)
    sq = host(eng.fp6_square_batch(dev(a)))
    mixed = np.concatenate([sq, a])
    wr, ws = ora.fp6_sqrt_batch(mixed)
    gr, gs = eng.fp6_sqrt_batch(dev(mixed))
    assert (host(gr) == wr).all() and (host(gs) == ws).all()
    assert ws[:n].all()
    hr, hs = eng.fp6_sqrt_batch(mixed)
    assert (hr == wr).all() and (hs == ws).all()
    kk = kats["fp3"]
    r, s_ = eng.fp6_sqrt_batch(np.array([kk["non_square_fp6"], [0] * 6, [1, 0, 0, 0, 0, 0], [P, P, P, P, P, P]], np.uint64))
    assert list(s_) == [0, 1, 1, 1] and not r[0].any() and not r[1].any() and list(r[2]) == [1, 0, 0, 0, 0, 0] and not r[3].any()
    # compressed round trip + rejection cases
    xy, inf = ora.mul_by_generator_batch(synth.scalars(62, n), 0)
    xy, inf = xy.copy(), inf.copy()
    if n > 8:
        xy[5] = [0] * 6 + [1] + [0] * 5
        inf[5] = 1
    enc = ora.encode_batch(xy, inf, True)
    for checked in (False, True):
        gxy, ginf, gsome = eng.from_compressed_batch(dev(enc), checked)
        assert (host(gxy) == xy).all() and (host(ginf) == inf).all() and host(gsome).all(), checked
    if n > 8:
        ns = np.array([kats["non_subgroup_point"]["x"] + kats["non_subgroup_point"]["y"]], np.uint64)
        enc[0, 0:8] = 0xFF
        enc[1, 48] |= 0x80
        enc[2] = 0
        enc[2, 48] = 0xC0
        enc[3] = ora.encode_batch(ns, np.zeros(1, np.uint8), True)[0]
        enc[4, 48] ^= 0x40
        enc[6] = 0      # x = 0 without the flag: B is a non-square (src/curve/encoding.rs:385-396)
        enc[7] = 0xFF   # src/curve/affine.rs:692-696
    for checked in (False, True):
        wxy, winf, wsome = ora.decode_compressed_batch(enc, checked)
        gxy, ginf, gsome = eng.from_compressed_batch(dev(enc), checked)
        assert (host(gxy) == wxy).all() and (host(ginf) == winf).all() and (host(gsome) == wsome).all(), checked
        hxy, hinf, hsome = eng.from_compressed_batch(enc, checked)
        assert (hxy == wxy).all() and (hinf == winf).all() and (hsome == wsome).all()
        if n > 8:
            assert list(wsome[:8]) == [0, 0, 0, 0 if checked else 1, 1, 1, 0, 0]
    # from_x
    xs = np.concatenate([xy[:, :6], np.array(kats["decompression"]["from_x_none"], np.uint64), np.zeros((1, 6), np.uint64), a])
    wxy, winf, wsome = ora.affine_from_x_batch(xs)
    gxy, ginf, gsome = eng.from_x_batch(dev(xs))
    assert (host(gxy) == wxy).all() and (host(ginf) == winf).all() and (host(gsome) == wsome).all()
    hxy, hinf, hsome = eng.from_x_batch(xs)
    assert (hxy == wxy).all() and (hinf == winf).all() and (hsome == wsome).all()
    assert wsome[n] == 0 and wsome[n + 1] == 1 and winf[n + 1] == 1


def test_compress_decompress_round_trip_large(eng):
    """Size-independent property at 2^18 points: k_i G -> 49 bytes -> from_compressed (checked) gives the points back."""
    import torch
    from cheetah_amd import AFFINE
    n = 1 << 18
    pts, inf = eng.mul_by_generator_batch(eng.synth_scalars(63, n), AFFINE)
    enc = eng.to_compressed_batch(pts, inf)
    xy, i2, some = eng.from_compressed_batch(enc, True)
    assert bool(some.all()) and torch.equal(xy.view(torch.int64), pts.view(torch.int64)) and torch.equal(i2, inf)


# ----------------------------------------- BASELINE-size property checks ----

def _golden_block_sums(name):
    """tests/golden/fullsize_<name>.json: per-2^20-block wrap-around sums of the ORACLE's output over the whole seeded
    input of a BASELINE configuration (tests/golden/generate_fullsize.py, CPU only)."""
    with open(os.path.join(ROOT, "tests", "golden", "fullsize_%s.json" % name)) as f:
        g = json.load(f)
    assert g["block"] == 1 << 20 and "oracle" in g["pinned_by"]
    return g


def _block_sums(t):
    """wrap-around 64-bit sums of the limbs of every 2^20-element block of a device array (rows = elements)"""
    import torch
    v = t.view(torch.int64)
    return ["%016x" % (int(x) & 0xFFFFFFFFFFFFFFFF) for x in v.reshape(v.shape[0] >> 20, -1).sum(dim=1).tolist()]


def test_full_size_configs_2_3_properties(eng, ora):
    """BASELINE configs 2 and 3 at full size (2^22): a strided sample is compared
    with the oracle, and the whole batch through size-independent properties:
    every k_i*G lies in the prime-order subgroup (q * P_i = O for ALL i), and
    ct == vartime for all i."""
    import torch
    from cheetah_amd import AFFINE
    n = 1 << 22
    S = eng.synth_scalars(2, n)
    pts, inf = eng.mul_by_generator_batch(S, AFFINE)             # config 2
    assert int(inf.sum()) == 0
    g2 = _golden_block_sums("config2")                           # ALL 2^22 results against the oracle's block sums
    assert _block_sums(pts) == g2["block_sums"] and g2["blocks"] == 4
    idx = np.arange(0, n, n // 64)
    sc_s = host(S)[idx]
    wxy, _ = ora.mul_by_generator_batch(sc_s, 0)
    assert (host(pts)[idx] == wxy).all()
    K = eng.synth_scalars(3, n)
    r_vt, i_vt = eng.multiply_vartime_batch(pts, K, AFFINE, AFFINE)  # config 3
    r_ct, i_ct = eng.multiply_batch(pts, K, AFFINE, AFFINE)
    assert torch.equal(r_vt.view(torch.int64), r_ct.view(torch.int64)) and torch.equal(i_vt, i_ct)
    g3 = _golden_block_sums("config3")
    assert _block_sums(r_vt) == g3["block_sums"] and int(i_vt.sum()) == sum(g3["identities_per_block"])
    wxy, winf = ora.multiply_batch(host(pts)[idx], 0, host(K)[idx], 0, vartime=True)
    assert (host(r_vt)[idx] == wxy).all() and (host(i_vt)[idx] == winf).all()
    qb = torch.from_numpy(np.frombuffer(Q.to_bytes(32, "little"), np.uint8).copy()).cuda().repeat(n, 1).contiguous()
    _, i_q = eng.multiply_vartime_batch(r_vt, qb, AFFINE, AFFINE)
    assert int(i_q.sum()) == n
    # config 3 as BASELINE.json words it: ProjectivePoint::multiply_vartime (every lane its own NAF sequence);
    # its (X:Y:Z), normalised, must be the same 2^22 affine points
    from cheetah_amd import PROJECTIVE
    pr, _ = eng.multiply_vartime_batch(pts, K, AFFINE, PROJECTIVE)
    nxy, ninf = eng.batch_normalize(pr, PROJECTIVE)
    assert torch.equal(nxy.view(torch.int64), r_ct.view(torch.int64)) and torch.equal(ninf, i_ct)
    didx = torch.from_numpy(idx).cuda()
    wpr, _ = ora.multiply_batch(host(take(pts, didx)), 0, host(K)[idx], 1, vartime=True)
    assert (host(take(pr, didx)) == wpr).all()


def test_full_size_config5_properties(eng, ora):
    """BASELINE config 5 at full size (2^26 seeded (P_i, k_i), bench.py's inputs), in four blocks of 2^24:
    the wrap-around sum of the canonical limbs of EVERY 2^20-block of results equals the sum the ORACLE produced for
    that block on the CPU (tests/golden/fullsize_config5.json, written by tests/golden/generate_fullsize.py: the oracle
    run over all 2^26 pairs, no GPU involved), constant-time == variable-time for ALL i (scan vs indexed lookups), a
    strided sample against the oracle, no identities; the sum of the block sums is the figure bench.py prints as
    `checksum` (tests/golden/config5_checksum.json)."""
    import torch
    from cheetah_amd import AFFINE
    m, blocks = 1 << 24, 4
    total = 0
    gold = _golden_block_sums("config5")
    assert gold["blocks"] == 64 and sum(gold["identities_per_block"]) == 0
    for c in range(blocks):
        r = eng.synth_scalars(5, m, (1 << 26) + c * m)
        pts, pinf = eng.mul_by_generator_batch(r, AFFINE)
        K = eng.synth_scalars(5, m, c * m)
        ct, ict = eng.multiply_batch(pts, K, AFFINE, AFFINE)
        vt, ivt = eng.multiply_vartime_batch(pts, K, AFFINE, AFFINE)
        assert torch.equal(ct.view(torch.int64), vt.view(torch.int64)) and torch.equal(ict, ivt)
        assert int(ict.sum()) == 0 and int(pinf.sum()) == 0
        total = (total + int(ct.view(torch.int64).sum().item())) & 0xFFFFFFFFFFFFFFFF
        sums = _block_sums(ct)
        bad = [16 * c + b for b in range(16) if sums[b] != gold["block_sums"][16 * c + b]]
        assert not bad, "2^20-blocks whose results differ from the oracle's: %r" % bad
        idx = torch.arange(c, m, m // 16, device="cuda")
        wxy, winf = ora.multiply_batch(host(take(pts, idx)), 0, host(take(K, idx)), 0)
        assert (host(take(ct, idx)) == wxy).all() and not winf.any()
        del r, pts, K, ct, vt
    assert "%016x" % total == gold["checksum"]
    with open(os.path.join(ROOT, "tests", "golden", "config5_checksum.json")) as f:
        assert "%016x" % total == json.load(f)["checksum"]


def test_full_size_config4_batch_normalize(eng, ora):
    """BASELINE config 4 at its full size, 2^24 Jacobian points (2.25 GiB): (x l^2, y l^3, l) built on the
    device from k_i*G, normalised back, must equal the affine points they came from, identities
    included; the first 4 096 also against the oracle's JacobianPoint::batch_normalize."""
    import torch
    from cheetah_amd import AFFINE, JACOBIAN
    n = 1 << 24
    S = eng.synth_scalars(4, n)
    aff, _ = eng.mul_by_generator_batch(S, AFFINE)
    lam = eng.synth_fp6(4, n)
    l2 = eng.fp6_square_batch(lam)
    l3 = eng.fp6_mul_batch(l2, lam)
    affi = aff.view(torch.int64)
    x = eng.fp6_mul_batch(affi[:, 0:6].contiguous(), l2)
    y = eng.fp6_mul_batch(affi[:, 6:12].contiguous(), l3)
    jac = torch.cat([x.view(torch.int64), y.view(torch.int64), lam.view(torch.int64)], dim=1).contiguous()
    ident = torch.arange(0, n, 1000, device="cuda")
    jac.view(torch.int64)[ident, 12:18] = 0
    xy, inf = eng.batch_normalize(jac, JACOBIAN)
    want = affi.clone()
    want[ident] = torch.tensor([0] * 6 + [1] + [0] * 5, dtype=torch.int64, device="cuda")
    assert torch.equal(xy.view(torch.int64), want.view(torch.int64))
    assert int(inf.sum()) == len(ident) and bool(inf[ident].all())
    g4 = _golden_block_sums("config4")                           # the oracle's batch_normalize over ALL 2^24 points, per block
    assert _block_sums(xy) == g4["block_sums"] and [int(x) for x in inf.view(n >> 20, -1).sum(dim=1).tolist()] == g4["identities_per_block"]
    idx = np.arange(0, 4096)
    wxy, winf = ora.batch_normalize(host(jac)[idx], True)
    assert (host(xy)[idx] == wxy).all() and (host(inf)[idx] == winf).all()


def test_cpp_host_api_on_gpu():
    """The C++ host API (cheetah_amd/host/cheetah.hpp) against the reference's KATs."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "host_api_test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_api_test: ok" in r.stdout
