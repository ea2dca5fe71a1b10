"""Pins the CPU oracle (oracle/cheetah_oracle.c) and the independent model
(oracle/model.py) against every known-answer vector the reference's own tests
hold for the hot path (tests/golden/reference_kats.json, each entry citing the
reference file:line), and against each other on seeded inputs."""
import ctypes as C

import numpy as np
import pytest

from cheetah_amd import synth
from oracle import model as M

P = M.P
Q = synth.Q


def u64(a):
    return np.asarray(a, dtype=np.uint64)


# ---------------------------------------------------------------- Fp KATs ----

def test_fp_kats(ora, kats):
    L = ora.lib()
    k = kats["fp"]
    largest = k["largest"]
    assert L.ora_fp_canonical(L.ora_fp_add(largest, largest)) == k["largest_plus_largest"]
    assert L.ora_fp_canonical(L.ora_fp_double(largest)) == k["largest_plus_largest"]
    assert L.ora_fp_canonical(L.ora_fp_add(largest, 1)) == k["largest_plus_one"]
    assert L.ora_fp_canonical(L.ora_fp_sub(largest, largest)) == 0
    assert L.ora_fp_canonical(L.ora_fp_sub(0, largest)) == L.ora_fp_canonical(L.ora_fp_sub(P, largest))
    assert L.ora_fp_canonical(L.ora_fp_neg(largest)) == k["neg_largest"]
    assert L.ora_fp_canonical(L.ora_fp_neg(0)) == 0
    assert L.ora_fp_canonical(L.ora_fp_neg(1)) == k["neg_one"]
    assert L.ora_reduce_u128(2**64 - 1, 2**64 - 1) % P == k["from_bytes_wide_max"]
    v = int.from_bytes(bytes(k["from_bytes_wide_neg_one_in"]), "little")
    assert L.ora_reduce_u128(v & (2**64 - 1), v >> 64) % P == P - 1
    a, b = k["sqrt_pair"]
    assert L.ora_fp_canonical(L.ora_fp_square(a)) == L.ora_fp_canonical(L.ora_fp_square(b))
    out = C.c_uint64()
    assert L.ora_fp_invert(0, C.byref(out)) == 0
    assert L.ora_fp_invert(1, C.byref(out)) == 1 and out.value % P == 1
    assert L.ora_fp_invert(P - 1, C.byref(out)) == 1 and out.value % P == P - 1


def test_fp_ops_vs_python(ora):
    L = ora.lib()
    rng = np.random.default_rng(7)
    vals = [0, 1, 2, P - 1, P, P + 1, 2**64 - 1, 2**32, 2**32 - 1, 2**63] + [int(v) for v in rng.integers(0, 2**64, 200, dtype=np.uint64)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        assert L.ora_fp_add(a, b) % P == (a + b) % P
        assert L.ora_fp_sub(a, b) % P == (a - b) % P
        assert L.ora_fp_mul(a, b) % P == (a * b) % P
        assert L.ora_fp_double(a) % P == (2 * a) % P
        assert L.ora_fp_triple(a) % P == (3 * a) % P
        assert L.ora_fp_neg(a) % P == (-a) % P
        assert L.ora_fp_mul_by_u32(a, b & 0xFFFFFFFF) % P == (a * (b & 0xFFFFFFFF)) % P
        out = C.c_uint64()
        if a % P:
            assert L.ora_fp_invert(a, C.byref(out)) == 1
            assert out.value % P == pow(a, P - 2, P)
            # test_invert_is_pow, src/fp.rs:914-934
            assert (out.value * a) % P == 1


# --------------------------------------------------------------- Fp6 KATs ----

def test_fp6_reference_kats(ora, kats):
    k = kats["fp6_mul"]
    assert list(ora.canon(ora.fp6_binop("ora_fp6_mul", k["a"], k["b"]))) == k["out"]
    k = kats["fp6_square"]
    assert list(ora.canon(ora.fp6_unop("ora_fp6_square", k["a"]))) == k["out"]
    k = kats["fp6_invert"]
    assert list(ora.canon(ora.fp6_invert(k["a"]))) == k["out"]
    assert list(ora.canon(ora.fp6_invert([1, 0, 0, 0, 0, 0]))) == [1, 0, 0, 0, 0, 0]
    assert ora.fp6_invert([0] * 6) is None
    assert ora.fp6_invert([P, 0, P, 0, 0, 0]) is None  # non-canonical zero
    # one * a == a  (src/fp6.rs:1582-1600)
    a = [1, 2, 3, 4, 5, 6]
    assert list(ora.canon(ora.fp6_binop("ora_fp6_mul", a, [1, 0, 0, 0, 0, 0]))) == a
    assert list(ora.canon(ora.fp6_binop("ora_fp6_add", a, [6, 5, 4, 3, 2, 1]))) == [7] * 6  # :1642-1668
    assert list(ora.canon(ora.fp6_binop("ora_fp6_sub", [6, 5, 4, 3, 2, 1], [3, 3, 2, 2, 1, 1]))) == [3, 2, 2, 1, 1, 0]  # :1672-1698
    # p^6 - 2 limbs (src/fp6.rs:1763-1770)
    e = sum(v << (64 * i) for i, v in enumerate(kats["p6_minus_2_limbs"]))
    assert e == P**6 - 2


def test_fp6_vs_model_seeded(ora):
    a = synth.fp6_elements(11, 64)
    b = synth.fp6_elements(12, 64)
    # a few non-canonical / edge operands
    a[0] = [2**64 - 1] * 6
    b[0] = [2**64 - 1] * 6
    a[1] = [P - 1] * 6
    b[1] = [P - 1] * 6
    a[2] = [0] * 6
    for i in range(64):
        ai, bi = M.f6(a[i]), M.f6(b[i])
        assert tuple(ora.canon(ora.fp6_binop("ora_fp6_mul", a[i], b[i]))) == M.f6_mul(ai, bi)
        assert tuple(ora.canon(ora.fp6_unop("ora_fp6_square", a[i]))) == M.f6_sqr(ai)
        assert tuple(ora.canon(ora.fp6_unop("ora_mul_by_3b", a[i]))) == M.f6_mul(ai, (1185, 3, 0, 0, 0, 0))
        assert tuple(ora.canon(ora.fp6_unop("ora_fp6_frobenius", a[i]))) == M.f6_frobenius(ai, 1)  # src/fp6.rs:1792-1802
        assert tuple(ora.canon(ora.fp6_unop("ora_fp6_frobenius_double", a[i]))) == M.f6_frobenius(ai, 2)
        assert tuple(ora.canon(ora.fp6_unop("ora_fp6_frobenius_triple", a[i]))) == M.f6_frobenius(ai, 3)
        if i < 8:
            inv = ora.fp6_invert(a[i])
            if ai == M.ZERO:
                assert inv is None
            else:
                assert tuple(ora.canon(inv)) == M.f6_inv(ai)  # test_invert_is_pow, src/fp6.rs:1760-1789


def test_fp6_golden_model_vectors(ora, model_vectors):
    for row in model_vectors["fp6"]:
        assert list(ora.canon(ora.fp6_binop("ora_fp6_mul", row["a"], row["b"]))) == row["mul"]
        assert list(ora.canon(ora.fp6_unop("ora_fp6_square", row["a"]))) == row["square"]
        assert list(ora.canon(ora.fp6_invert(row["a"]))) == row["invert"]


# ------------------------------------------------------------- recoders ----

def test_recoders_reconstruct(ora):
    # src/scalar.rs:1466-1534 are round trips: digits reconstruct the scalar.
    L = ora.lib()
    sc = list(synth.scalars(5, 40)) + [np.zeros(32, np.uint8), np.full(32, 255, np.uint8),
                                       np.frombuffer((M.Q - 1).to_bytes(32, "little"), np.uint8).copy()]
    for s in sc:
        s = np.ascontiguousarray(s)
        k = int.from_bytes(bytes(s), "little")
        d16 = np.zeros(64, np.int8)
        L.ora_bytes_to_radix_16(s, d16)
        assert sum(int(d) << (4 * i) for i, d in enumerate(d16)) == k
        assert all(-8 <= int(d) < 8 for d in d16[:63]) and 0 <= int(d16[63]) <= 16
        d256 = np.zeros(32, np.int8)
        L.ora_bytes_to_radix_256(s, d256)
        if s[31] < 127:  # the top digit is not recentred and wraps as i8 above that
            assert sum(int(d) << (8 * i) for i, d in enumerate(d256)) == k
        for w in (5, 8):
            naf = np.zeros(256, np.int8)
            L.ora_bytes_to_wnaf_vartime(s, w, naf)
            if k < 2**255:  # a set bit 255 can leave a final carry outside the 256 digits
                assert sum(int(d) << i for i, d in enumerate(naf)) == k
            nz = [i for i, d in enumerate(naf) if d]
            assert all(int(naf[i]) % 2 == 1 and abs(int(naf[i])) < 2 ** (w - 1) for i in nz)
            assert all(b - a >= w for a, b in zip(nz, nz[1:]))


# ------------------------------------------------------------ curve KATs ----

def _aff(d):
    return u64(d["x"] + d["y"])


def _proj_of(xy):
    return np.concatenate([u64(xy), u64([1, 0, 0, 0, 0, 0])])


def test_curve_reference_kats(ora, kats):
    L = ora.lib()
    g = _aff(kats["generator"])
    assert list(ora.table("ora_generator_affine", 1)[0]) == list(g)
    assert list(ora.table("ora_shift_point_affine", 1)[0]) == list(_aff(kats["shift_point"]))
    gp = _proj_of(g)
    assert L.ora_proj_is_on_curve(gp) == 1 and L.ora_jac_is_on_curve(gp) == 1
    two_g = _aff(kats["two_g_affine"])
    xy = np.zeros(12, np.uint64)
    inf = np.zeros(1, np.uint8)
    for dbl, to_aff in (("ora_proj_double", "ora_proj_to_affine"), ("ora_proj_double_unchecked", "ora_proj_to_affine"),
                        ("ora_jac_double", "ora_jac_to_affine"), ("ora_jac_double_unchecked", "ora_jac_to_affine")):
        out = np.zeros(18, np.uint64)
        getattr(L, dbl)(gp, out)
        getattr(L, to_aff)(out, xy, inf)
        assert list(xy) == list(two_g) and inf[0] == 0, dbl
    # complete addition G + G and mixed addition give 2G too
    out = np.zeros(18, np.uint64)
    L.ora_proj_add(gp, gp, out)
    L.ora_proj_to_affine(out, xy, inf)
    assert list(xy) == list(two_g)
    L.ora_proj_add_mixed(gp, g, 0, out)
    L.ora_proj_to_affine(out, xy, inf)
    assert list(xy) == list(two_g)
    # identity handling (src/curve/projective.rs:1145-1150, :1185-1190)
    ident = u64([0] * 6 + [1] + [0] * 5 + [0] * 6)
    L.ora_proj_double(ident, out)
    L.ora_proj_to_affine(out, xy, inf)
    assert inf[0] == 1 and list(xy) == [0] * 6 + [1] + [0] * 5
    L.ora_proj_add(ident, gp, out)
    L.ora_proj_to_affine(out, xy, inf)
    assert list(xy) == list(g)
    # the non-subgroup point is on the curve (src/curve/projective.rs:1427)
    ns = _proj_of(_aff(kats["non_subgroup_point"]))
    assert L.ora_proj_is_on_curve(ns) == 1
    # byte strings
    assert int.from_bytes(bytes(kats["fq_modulus_bytes"]), "little") == M.Q
    assert int.from_bytes(bytes(kats["cofactor_bytes"]), "little") == M.COFACTOR


def test_is_torsion_free_ladder(ora, kats):
    """q * G = O through the full vartime ladder (src/curve/projective.rs:1462)
    and the non-subgroup point is not torsion free (:1460); cofactor clearing
    brings it into the subgroup (:1429-1433)."""
    L = ora.lib()
    qb = np.array(kats["fq_modulus_bytes"], np.uint8)
    hb = np.array(kats["cofactor_bytes"], np.uint8)
    gp = _proj_of(_aff(kats["generator"]))
    out = np.zeros(18, np.uint64)
    L.ora_proj_multiply_vartime(gp, qb, out)
    assert L.ora_fp6_is_zero(np.ascontiguousarray(out[12:18])) == 1
    L.ora_jac_multiply_vartime(gp, qb, out)
    assert L.ora_fp6_is_zero(np.ascontiguousarray(out[12:18])) == 1
    ns = _proj_of(_aff(kats["non_subgroup_point"]))
    L.ora_proj_multiply_vartime(ns, qb, out)
    assert L.ora_fp6_is_zero(np.ascontiguousarray(out[12:18])) == 0
    cleared = np.zeros(18, np.uint64)
    L.ora_proj_multiply_vartime(ns, hb, cleared)
    assert L.ora_proj_is_on_curve(cleared) == 1
    L.ora_proj_multiply_vartime(cleared, qb, out)
    assert L.ora_fp6_is_zero(np.ascontiguousarray(out[12:18])) == 1


def test_tables_vs_model(ora, kats):
    g = (tuple(kats["generator"]["x"]), tuple(kats["generator"]["y"]))
    s = (tuple(kats["shift_point"]["x"]), tuple(kats["shift_point"]["y"]))
    ms = ora.table("ora_minus_shift_point_array", 257)
    acc = M.aff_neg(s)
    for i in range(257):
        if i in (0, 1, 2, 3, 4, 5, 100, 255, 256):
            assert tuple(int(v) for v in ms[i]) == acc[0] + acc[1], i
        acc = M.aff_add(acc, acc)
    bpt = ora.table("ora_basepoint_table", 256)
    for i in (0, 1, 7, 31):
        base = M.scalar_mul(256**i, g)
        for j in (0, 1, 7):
            pt = M.scalar_mul(j + 1, base)
            assert tuple(int(v) for v in bpt[8 * i + j]) == pt[0] + pt[1], (i, j)
    bpl = ora.table("ora_basepoint_lookup", 128)
    for j in (0, 1, 2, 64, 127):
        pt = M.scalar_mul(j + 1, g)
        assert tuple(int(v) for v in bpl[j]) == pt[0] + pt[1]


# -------------------------------------------- ladders: cross-path + model ----

def test_ladders_cross_path_and_model(ora, model_vectors):
    """multiply == multiply_vartime == BASEPOINT_TABLE.multiply == BASEPOINT_LOOKUP
    (src/curve/projective.rs:1328-1351, src/curve/jacobian.rs:1612-1626), all equal
    to the independent model's k*G."""
    rows = model_vectors["mul_by_generator"]
    sc = np.array([r["scalar"] for r in rows], np.uint8)
    n = len(sc)
    g = ora.table("ora_generator_affine", 1)[0]
    gs = np.tile(g, (n, 1))
    want_xy = [r["out"]["xy"] for r in rows]
    want_inf = [r["out"]["inf"] for r in rows]
    canonical = [int.from_bytes(bytes(s), "little") < 2**255 for s in sc]

    xy, inf = ora.mul_by_generator_batch(sc, 0)
    for i in range(n):
        if sc[i][31] >> 4 <= 7:  # BasePointTable covers the top digit only then
            assert list(xy[i]) == want_xy[i] and inf[i] == want_inf[i], i
    xy, inf = ora.multiply_batch(gs, 0, sc, 0)
    assert [list(r) for r in xy] == want_xy and list(inf) == want_inf
    xy, inf = ora.multiply_batch(gs, 0, sc, 0, vartime=True)
    for i in range(n):
        if canonical[i]:
            assert list(xy[i]) == want_xy[i] and inf[i] == want_inf[i], i
    L = ora.lib()
    for i in range(n):
        if sc[i][31] < 127:
            out = np.zeros(18, np.uint64)
            L.ora_basepoint_lookup_multiply(np.ascontiguousarray(sc[i]), out)
            a = M.jac_to_affine([int(v) for v in out])
            assert M.affine_to_limbs(a) == (want_xy[i], want_inf[i]), i
            gp = np.concatenate([g, u64([1, 0, 0, 0, 0, 0])])
            L.ora_jac_multiply(gp, np.ascontiguousarray(sc[i]), out)
            assert M.affine_to_limbs(M.jac_to_affine([int(v) for v in out])) == (want_xy[i], want_inf[i]), i
            L.ora_jac_multiply_vartime(gp, np.ascontiguousarray(sc[i]), out)
            assert M.affine_to_limbs(M.jac_to_affine([int(v) for v in out])) == (want_xy[i], want_inf[i]), i


def test_variable_base_vs_model(ora, model_vectors):
    rows = model_vectors["multiply"]
    pts = np.array([r["point"]["xy"] for r in rows], np.uint64)
    sc = np.array([r["scalar"] for r in rows], np.uint8)
    xy, inf = ora.multiply_batch(pts, 0, sc, 0)
    for i, r in enumerate(rows):
        assert list(xy[i]) == r["out"]["xy"] and inf[i] == r["out"]["inf"], i
    xy, inf = ora.multiply_batch(pts, 0, sc, 0, vartime=True)
    for i, r in enumerate(rows):
        if int.from_bytes(bytes(sc[i]), "little") < 2**255:
            assert list(xy[i]) == r["out"]["xy"] and inf[i] == r["out"]["inf"], i
    # projective output normalises to the same affine point
    pr, _ = ora.multiply_batch(pts, 0, sc, 1)
    for i, r in enumerate(rows):
        assert M.affine_to_limbs(M.proj_to_affine([int(v) for v in pr[i]])) == (r["out"]["xy"], r["out"]["inf"])


def test_g_a_b_equals_g_ab(ora):
    # (g*a)*b == g*(a*b)   src/curve/projective.rs:1338
    a = synth.scalars(21, 4)
    b = synth.scalars(22, 4)
    ga, _ = ora.mul_by_generator_batch(a, 0)
    gab, inf1 = ora.multiply_batch(ga, 0, b, 0)
    ab = np.array([list(((int.from_bytes(bytes(x), "little") * int.from_bytes(bytes(y), "little")) % M.Q).to_bytes(32, "little"))
                   for x, y in zip(a, b)], np.uint8)
    want, inf2 = ora.mul_by_generator_batch(ab, 0)
    assert (gab == want).all() and (inf1 == inf2).all()


# ------------------------------------------------------ batch_normalize ----

def test_batch_normalize_vs_model_and_identities(ora, model_vectors):
    rows = model_vectors["normalize"]
    jac = np.array([r["jacobian"] for r in rows], np.uint64)
    prj = np.array([r["projective"] for r in rows], np.uint64)
    want = np.array([r["affine"]["xy"] for r in rows], np.uint64)
    for chunk in (0, 1, 3):
        xy, inf = ora.batch_normalize(jac, True, chunk)
        assert (xy == want).all() and not inf.any()
        xy, inf = ora.batch_normalize(prj, False, chunk)
        assert (xy == want).all() and not inf.any()
    # identities (the branch the reference's own test never takes):
    # every subset of {0, 3, 7} replaced by z = 0 (x, y arbitrary)
    ident_xy = [0] * 6 + [1] + [0] * 5
    for mask in range(8):
        j2, p2, w2 = jac.copy(), prj.copy(), want.copy()
        winf = np.zeros(len(rows), np.uint8)
        for bit, idx in enumerate((0, 3, 7)):
            if mask >> bit & 1:
                j2[idx, 12:18] = 0
                p2[idx, 12:18] = [P, 0, 0, P, 0, 0]  # non-canonical zero
                w2[idx] = ident_xy
                winf[idx] = 1
        xy, inf = ora.batch_normalize(j2, True)
        assert (xy == w2).all() and (inf == winf).all()
        xy, inf = ora.batch_normalize(p2, False)
        assert (xy == w2).all() and (inf == winf).all()
    # per-point conversion agrees (src/curve/projective.rs:1466-1502)
    L = ora.lib()
    xy1 = np.zeros(12, np.uint64)
    i1 = np.zeros(1, np.uint8)
    for i in range(len(rows)):
        L.ora_jac_to_affine(np.ascontiguousarray(jac[i]), xy1, i1)
        assert (xy1 == want[i]).all()
        L.ora_proj_to_affine(np.ascontiguousarray(prj[i]), xy1, i1)
        assert (xy1 == want[i]).all()
    # empty input
    xy, inf = ora.batch_normalize(np.zeros((0, 18), np.uint64), True)
    assert xy.shape == (0, 12)


def test_encodings_kats_and_round_trip(ora, kats):
    """Point encodings (src/curve/encoding.rs): the reference's lexicographic KATs (src/fp6.rs:1805-1845),
    identity byte strings (src/curve/projective.rs:1533-1539, :1575-1582), the (0, 0) point rejected by
    from_uncompressed (:1587-1597), and encode -> decode round trips on seeded subgroup points."""
    L = ora.lib()
    for c in kats["fp6_lexicographically_largest"]["cases"]:
        assert L.ora_fp6_lexicographically_largest(np.array(c["a"], np.uint64)) == c["largest"]
    # the expression is not a strict lexicographic order: c5 small and non-zero, c4 = 0, c3 large -> "largest"
    assert L.ora_fp6_lexicographically_largest(np.array([0, 0, 0, P - 1, 0, 1], np.uint64)) == 1
    ident = np.array([[0] * 6 + [1] + [0] * 5], np.uint64)
    one = np.array([1], np.uint8)
    assert list(ora.encode_batch(ident, one, True)[0]) == kats["identity_encodings"]["compressed"]
    assert list(ora.encode_batch(ident, one, False)[0]) == kats["identity_encodings"]["uncompressed"]
    n = 40
    xy, inf = ora.mul_by_generator_batch(synth.scalars(21, n), 0)
    unc = ora.encode_batch(xy, inf, False)
    cmp_ = ora.encode_batch(xy, inf, True)
    assert (unc[:, :48] == cmp_[:, :48]).all() and (unc[:, 96] == 0).all()
    for i in range(n):
        y = [int(v) for v in xy[i, 6:]]
        neg = [(P - v) % P for v in y]
        bit = (cmp_[i, 48] >> 6) & 1
        # exactly one of y, -y carries the flag (their c5 are non-zero negatives of each other)
        assert bit == L.ora_fp6_lexicographically_largest(np.array(y, np.uint64))
        assert bit != L.ora_fp6_lexicographically_largest(np.array(neg, np.uint64))
    for checked in (0, 1):
        dxy, dinf, some = ora.decode_uncompressed_batch(unc, checked)
        assert some.all() and (dxy == xy).all() and (dinf == inf).all()
    # invalid inputs: non-canonical limb; (0, 0) without the flag (unchecked accepts it, checked does not);
    # infinity flag with non-zero coordinates; a point on the curve outside the subgroup
    bad = unc[:4].copy()
    bad[0, 0:8] = 0xFF
    bad[1, :] = 0
    bad[2, 96] = 0x80
    ns = np.array([kats["non_subgroup_point"]["x"] + kats["non_subgroup_point"]["y"]], np.uint64)
    bad[3] = ora.encode_batch(ns, np.zeros(1, np.uint8), False)[0]
    _, _, some0 = ora.decode_uncompressed_batch(bad, 0)
    _, _, some1 = ora.decode_uncompressed_batch(bad, 1)
    assert list(some0) == [0, 1, 0, 1] and list(some1) == [0, 0, 0, 0]
    dxy, dinf, some = ora.decode_uncompressed_batch(np.array([kats["identity_encodings"]["uncompressed"]], np.uint8), 1)
    assert some[0] == 1 and dinf[0] == 1 and list(dxy[0]) == [0] * 6 + [1] + [0] * 5


def test_multi_scalar_ladders_vs_single(ora):
    """multiply_double / multiply_many (+ vartime, + with_basepoint) against sums of single products
    (the relation the reference tests at src/curve/projective.rs:1354-1397)."""
    n, m = 6, 3
    k = synth.scalars(31, n * m)
    r = synth.scalars(32, n)
    base, _ = ora.mul_by_generator_batch(r, 0)
    pts = np.repeat(base, m, axis=0)  # every group multiplies the same point: sum_j k_j P = (sum_j k_j) P
    ksum = np.array([list((sum(int.from_bytes(bytes(k[g * m + j]), "little") for j in range(m)) % Q).to_bytes(32, "little"))
                     for g in range(n)], np.uint8)
    want, winf = ora.multiply_batch(base, 0, ksum, 0)
    for vt in (False, True):
        for out_form in (0, 1, 2):
            got, ginf = ora.multiply_many_batch(pts, 0, k, m, out_form, vartime=vt)
            if out_form:
                got, ginf = ora.batch_normalize(got, out_form == 2)
            assert (got == want).all() and (ginf == winf).all(), (vt, out_form)
    # a * P + b * G with P = r * G  ==  (a r + b) * G
    ab = synth.scalars(33, 2 * n)
    got, _ = ora.multiply_many_batch(base, 0, ab, 1, 0, vartime=True, with_basepoint=True)
    comb = np.array([list(((int.from_bytes(bytes(ab[2 * i]), "little") * int.from_bytes(bytes(r[i]), "little")
                            + int.from_bytes(bytes(ab[2 * i + 1]), "little")) % Q).to_bytes(32, "little")) for i in range(n)], np.uint8)
    assert (got == ora.mul_by_generator_batch(comb, 0)[0]).all()


def _emb3(a):
    """Fp3 element (a0, a1, a2) embedded in Fp6 as (c0, c2, c4) (src/fp3.rs:112-120)."""
    return (int(a[0]) % P, 0, int(a[1]) % P, 0, int(a[2]) % P, 0)


def test_fp3_and_sqrt_kats(ora, kats):
    """Fp3 mul / square KATs (src/fp3.rs:437-450, :477-509), the 2^32-th root of unity (:515-525), the
    non-squares of Fp3 and Fp6 (src/fp3.rs:463-473, src/fp6.rs:1564-1577), sqrt(0) = 0, sqrt(1) = 1, and
    seeded squares / non-squares against the big-integer model (root^2 = a; Euler's criterion)."""
    L = ora.lib()
    k = kats["fp3"]
    o = np.zeros(3, np.uint64)
    L.ora_fp3_mul(u64(k["mul"]["a"]), u64(k["mul"]["b"]), o)
    assert list(o) == k["mul"]["out"]
    L.ora_fp3_mul(u64(k["square"]["a"]), u64(k["square"]["a"]), o)
    assert list(o) == k["square"]["out"]
    L.ora_fp3_mul(u64([1, 3, 5]), u64([1, 0, 0]), o)
    assert list(o) == [1, 3, 5]
    z = k["two_adic_root_of_unity"]
    assert pow(z, 1 << 32, P) == 1 and pow(z, (1 << 32) - 1, P) != 1 and pow(z, 1 << 31, P) == P - 1
    assert L.ora_fp3_sqrt(u64(k["non_square_fp3"]), o) == 0
    assert L.ora_fp3_sqrt(u64([0, 0, 0]), o) == 1 and list(o) == [0, 0, 0]
    assert L.ora_fp3_sqrt(u64([1, 0, 0]), o) == 1 and list(o) == [1, 0, 0]
    rng = np.random.default_rng(5)
    for _ in range(20):
        a = [int(v) % P for v in rng.integers(0, 2**64, 3, dtype=np.uint64)]
        b = [int(v) % P for v in rng.integers(0, 2**64, 3, dtype=np.uint64)]
        L.ora_fp3_mul(u64(a), u64(b), o)
        want = M.f6_mul(_emb3(a), _emb3(b))
        assert _emb3(o) == want
        sq = M.f6_sqr(_emb3(a))
        assert L.ora_fp3_sqrt(u64([sq[0], sq[2], sq[4]]), o) == 1
        assert M.f6_sqr(_emb3(o)) == sq
    o6 = np.zeros(6, np.uint64)
    assert L.ora_fp6_sqrt(u64(k["non_square_fp6"]), o6) == 0 and not o6.any()
    assert L.ora_fp6_sqrt(u64([0] * 6), o6) == 1 and list(o6) == [0] * 6
    assert L.ora_fp6_sqrt(u64([1, 0, 0, 0, 0, 0]), o6) == 1 and list(o6) == [1, 0, 0, 0, 0, 0]
    a = synth.fp6_elements(41, 24)
    sq = np.array([M.f6_sqr(M.f6(r)) for r in a[:12]], np.uint64)
    mixed = np.concatenate([sq, a[12:]])
    roots, some = ora.fp6_sqrt_batch(mixed)
    assert some[:12].all()
    euler = (P ** 6 - 1) // 2
    for i in range(len(mixed)):
        is_sq = M.f6_pow(M.f6(mixed[i]), euler) == M.ONE
        assert bool(some[i]) == is_sq, i
        if is_sq:
            assert M.f6_sqr(M.f6(roots[i])) == M.f6(mixed[i])
        else:
            assert not roots[i].any()


def test_decompression_and_from_x(ora, kats):
    """CompressedPoint::to_affine / AffinePoint::from_compressed (src/curve/encoding.rs:82-129,
    src/curve/affine.rs:303-309) and AffinePoint::from_x (:326-343): round trips on seeded subgroup points
    (the reference's test shape, src/curve/encoding.rs:359-397, src/curve/affine.rs:627-697), its rejection KATs,
    the non-subgroup point (accepted unchecked, rejected checked), and the recovered y checked on the curve by the model."""
    n = 24
    xy, inf = ora.mul_by_generator_batch(synth.scalars(51, n), 0)
    xy, inf = xy.copy(), inf.copy()
    xy[5] = [0] * 6 + [1] + [0] * 5
    inf[5] = 1
    enc = ora.encode_batch(xy, inf, True)
    for checked in (0, 1):
        dxy, dinf, some = ora.decode_compressed_batch(enc, checked)
        assert some.all() and (dxy == xy).all() and (dinf == inf).all()
    d = kats["decompression"]
    bad = np.array(d["compressed_none"], np.uint8)
    for checked in (0, 1):
        dxy, dinf, some = ora.decode_compressed_batch(bad, checked)
        assert not some.any() and not dxy.any() and not dinf.any()
    ns = np.array([kats["non_subgroup_point"]["x"] + kats["non_subgroup_point"]["y"]], np.uint64)
    cases = enc[:5].copy()
    cases[0, 0:8] = 0xFF            # non-canonical limb
    cases[1, 48] |= 0x80            # infinity flag over a non-zero x
    cases[2] = 0
    cases[2, 48] = 0xC0             # infinity flag and sort flag
    cases[3] = ora.encode_batch(ns, np.zeros(1, np.uint8), True)[0]
    cases[4, 48] ^= 0x40            # other root: the negated point, still in the subgroup
    dxy0, _, some0 = ora.decode_compressed_batch(cases, 0)
    dxy1, _, some1 = ora.decode_compressed_batch(cases, 1)
    assert list(some0) == [0, 0, 0, 1, 1] and list(some1) == [0, 0, 0, 0, 1]
    assert (dxy0[3] == ns[0]).all()
    assert list(dxy0[4, :6]) == list(xy[4, :6]) and [int(v) for v in dxy0[4, 6:]] == [(P - int(v)) % P for v in xy[4, 6:]]
    # from_x
    xs = np.concatenate([xy[:, :6], np.array(d["from_x_none"], np.uint64), np.zeros((1, 6), np.uint64), synth.fp6_elements(52, 8)])
    fxy, finf, fsome = ora.affine_from_x_batch(xs)
    L = ora.lib()
    for i in range(n):
        if i == 5:
            continue
        assert fsome[i] and (fxy[i, :6] == xy[i, :6]).all()
        y = [int(v) for v in xy[i, 6:]]
        want = y if L.ora_fp6_lexicographically_largest(np.array(y, np.uint64)) else [(P - v) % P for v in y]
        assert [int(v) for v in fxy[i, 6:]] == want
    assert fsome[5] and finf[5] == 1 and list(fxy[5]) == [0] * 6 + [1] + [0] * 5     # x = 0: the identity
    assert fsome[n] == 0 and not fxy[n].any()                                          # from_x(1): none
    assert fsome[n + 1] == 1 and finf[n + 1] == 1
    for i in range(n + 2, len(xs)):                                                    # arbitrary x: on the curve or none
        if fsome[i]:
            pt = (M.f6(fxy[i, :6]), M.f6(fxy[i, 6:]))
            assert M.is_on_curve(pt) and L.ora_fp6_lexicographically_largest(np.ascontiguousarray(fxy[i, 6:])) == 1
        else:
            rhs = M.f6_add(M.f6_add(M.f6_mul(M.f6_sqr(M.f6(xs[i])), M.f6(xs[i])), M.f6(xs[i])), M.B)
            assert M.f6_pow(rhs, (P ** 6 - 1) // 2) != M.ONE
    assert 0 < fsome[n + 2:].sum() < 8


# ------------------------------------------- complete group law (oracle) ----

def _rescale(xy, z, jac):
    """(x, y) -> the representative with the given z: (x z, y z, z) or (x z^2, y z^3, z)."""
    x, y = tuple(int(v) for v in xy[:6]), tuple(int(v) for v in xy[6:])
    z = tuple(int(v) % P for v in z)
    if jac:
        z2 = M.f6_sqr(z)
        return u64(M.f6_mul(x, z2) + M.f6_mul(y, M.f6_mul(z2, z)) + z)
    return u64(M.f6_mul(x, z) + M.f6_mul(y, z) + z)


def test_complete_additions_mirror_reference_tests(ora, kats):
    """JacobianPoint::add / add_mixed / double and their projective counterparts, after
    src/curve/jacobian.rs:1405-1609 and src/curve/projective.rs tests of the same names."""
    L = ora.lib()
    g = _aff(kats["generator"])
    gj = _proj_of(g)
    z = [0xa3c62f9770336022, 0x69a1531152c92fc1, 0x1636d9b90656a08a, 0x635289066593aaf6, 0x1e2178e3c54f6682,
         0xe15458e6d847f393]  # src/curve/jacobian.rs:1455-1462
    out, out2 = np.zeros(18, np.uint64), np.zeros(18, np.uint64)
    jid = u64([1] + [0] * 5 + [1] + [0] * 5 + [0] * 6)     # JacobianPoint::identity(), src/curve/jacobian.rs:223-229
    pid = u64([0] * 6 + [1] + [0] * 5 + [0] * 6)           # ProjectivePoint::identity(), src/curve/projective.rs:204-210
    aid = u64([0] * 6 + [1] + [0] * 5)
    # identity + identity, identity.double()
    L.ora_jac_add(jid, jid, out)
    assert list(out) == list(jid)
    L.ora_jac_double(jid, out)
    assert list(out) == list(jid)
    L.ora_jac_add_mixed(jid, aid, 1, out)
    assert list(out) == list(jid)
    # identity + b, b + identity with b a rescaled generator: equal to the generator as points, b itself as coordinates
    b = _rescale(g, z, True)
    assert L.ora_jac_is_on_curve(b) == 1 and L.ora_jac_eq(b, gj) == 1 and L.ora_jac_eq(b, jid) == 0
    L.ora_jac_add(jid, b, out)
    assert list(out) == list(b)
    L.ora_jac_add(b, jid, out)
    assert list(out) == list(b)
    L.ora_jac_add_mixed(b, aid, 1, out)
    assert list(out) == list(b)
    # equality is a relation on points: the negated representative differs, a broken x is off the curve
    c = b.copy()
    c[6:12] = (P - c[6:12]) % P
    assert L.ora_jac_is_on_curve(c) == 1 and L.ora_jac_eq(c, gj) == 0
    # 4G + 2G == G + 5 x G (add), == G + 5 x affine G (add_mixed)
    two, four = np.zeros(18, np.uint64), np.zeros(18, np.uint64)
    L.ora_jac_double(gj, two)
    L.ora_jac_double(two, four)
    L.ora_jac_add(four, two, out)
    d, d2 = gj.copy(), gj.copy()
    for _ in range(5):
        L.ora_jac_add(d.copy(), gj, d)
        L.ora_jac_add_mixed(d2.copy(), g, 0, d2)
    assert L.ora_jac_is_on_curve(out) == 1 and L.ora_jac_eq(out, d) == 1 and L.ora_jac_eq(out, d2) == 1
    six = M.scalar_mul(6, (tuple(kats["generator"]["x"]), tuple(kats["generator"]["y"])))
    assert M.jac_to_affine([int(v) for v in out]) == six
    # a + a takes the doubling branch: the same representative as a.double(); so does add_mixed with a's own affine form
    L.ora_jac_add(b, b, out)
    L.ora_jac_double(b, out2)
    assert list(out) == list(out2)
    L.ora_jac_add_mixed(b, g, 0, out)
    assert list(out) == list(out2)
    # ... which the unchecked formula cannot do (src/curve/jacobian.rs:1598-1599)
    L.ora_jac_add_mixed_unchecked(b, g, 0, out)
    assert L.ora_jac_eq(out, out2) == 0
    # a + (-a) == identity, exactly (1, 1, 0) (src/curve/jacobian.rs:1605-1609)
    L.ora_jac_add(two, u64(list(two[:6]) + [(P - int(v)) % P for v in two[6:12]] + list(two[12:])), out)
    assert list(out) == list(jid)
    neg_g = u64(list(g[:6]) + [(P - int(v)) % P for v in g[6:]])
    L.ora_jac_add_mixed(b, neg_g, 0, out)
    assert list(out) == list(jid)
    # projective: equality, mixed addition with the identity, 4G + 2G
    bp = _rescale(g, z, False)
    gp = _proj_of(g)
    assert L.ora_proj_is_on_curve(bp) == 1 and L.ora_proj_eq(bp, gp) == 1 and L.ora_proj_eq(bp, pid) == 0
    L.ora_proj_add_mixed(bp, aid, 1, out)
    assert list(out) == list(bp)
    L.ora_proj_double(gp, two)
    L.ora_proj_double(two, four)
    L.ora_proj_add(four, two, out)
    assert M.proj_to_affine([int(v) for v in out]) == six
    d = gp.copy()
    for _ in range(5):
        L.ora_proj_add_mixed(d.copy(), g, 0, d)
    assert L.ora_proj_eq(out, d) == 1


def test_complete_additions_vs_affine_model(ora):
    """Seeded points in random representatives: every complete formula lands on the model's affine chord-and-tangent
    sum; add_mixed equals add_mixed_unchecked away from the special cases (src/curve/jacobian.rs:1589-1596)."""
    n = 24
    xy, _, ok = ora.affine_from_x_batch(synth.fp6_elements(71, 8 * n))
    pts = xy[ok == 1][: 2 * n]
    zs = synth.fp6_elements(72, 2 * n).reshape(-1, 6)
    for jac in (False, True):
        form = 2 if jac else 1
        lhs = np.stack([_rescale(pts[i], zs[i], jac) for i in range(n)])
        rhs = np.stack([_rescale(pts[n + i], zs[n + i], jac) for i in range(n)])
        s = ora.add_batch(lhs, form, rhs, form)
        sm = ora.add_batch(lhs, form, pts[n:], 0)
        dd = ora.double_batch(lhs, form)
        to_aff = M.jac_to_affine if jac else M.proj_to_affine
        for i in range(n):
            p1, p2 = M.limbs_to_affine([int(v) for v in pts[i]]), M.limbs_to_affine([int(v) for v in pts[n + i]])
            want = M.aff_add(p1, p2)
            assert to_aff([int(v) for v in s[i]]) == want
            assert to_aff([int(v) for v in sm[i]]) == want
            assert to_aff([int(v) for v in dd[i]]) == M.aff_add(p1, p1)
        unchecked = np.zeros(18, np.uint64)
        f = ora.lib().ora_jac_add_mixed_unchecked if jac else ora.lib().ora_proj_add_mixed_unchecked
        eq = ora.lib().ora_jac_eq if jac else ora.lib().ora_proj_eq
        for i in range(n):
            f(lhs[i], pts[n + i], 0, unchecked)
            if jac:
                assert list(unchecked % np.uint64(P)) == list(sm[i])   # the same sequence, hence the same representative
            else:
                assert eq(unchecked, sm[i]) == 1                       # another formula: the same point
