"""GPU tests of the boundary itself (include/cheetah_hip.h), through the C ABI:
  * the multi-device context: creation contract, error paths, and -- on this
    one-GPU box -- its sharding / peer-copy / gather paths with all shards on one device
    (cheetah_debug_ctx_create_replicated), every result equal to the single-device one;
  * the host-pointer pipeline: batches of several chunks equal the `_dev` results;
  * BasePointTable::create(P) / multiply for caller-supplied bases against the oracle (L1 + L2),
    and the reference's own relation BasePointTable::create(g).multiply == g * s
    (src/curve/projective.rs:1328-1351);
  * the out-of-range rule of the BASEPOINT_TABLE forms; is_on_curve for the three point types.
Bar: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from cheetah_amd import _capi, synth

pytestmark = pytest.mark.gpu

AFFINE, PROJECTIVE, JACOBIAN = 0, 1, 2
P = 0xFFFFFFFF00000001
IDENT = np.array([0] * 6 + [1] + [0] * 5, np.uint64)


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cheetah_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng3():
    """Three shards, all on device 0."""
    from cheetah_amd import Engine
    e = Engine(0, replicate=3)
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


def points(eng, seed, n):
    """n affine points r_i G (host arrays), a few identities mixed in."""
    xy, inf = eng.mul_by_generator_batch(synth.scalars(seed, n, 1 << 22), AFFINE)
    xy = xy.copy()
    for i in range(3, n, 97):
        xy[i] = IDENT
    return xy


# ---------------------------------------------------------------- context -----

def test_ctx_create_contract():
    import torch
    lib = _capi.load()
    ndev = torch.cuda.device_count()
    h = C.c_void_p()

    def create(ids):
        return lib.cheetah_ctx_create((C.c_int * max(1, len(ids)))(*ids), len(ids), C.byref(h))

    assert create([]) == _capi.ERR_INVALID and not h.value
    assert lib.cheetah_ctx_create(None, 1, C.byref(h)) == _capi.ERR_INVALID
    assert lib.cheetah_ctx_create((C.c_int * 1)(0), 1, None) == _capi.ERR_INVALID
    assert create([ndev]) == _capi.ERR_INVALID and b"out of range" in lib.cheetah_last_error(None)
    assert create([-1]) == _capi.ERR_INVALID
    assert create([0, 0]) == _capi.ERR_INVALID and b"twice" in lib.cheetah_last_error(None)
    assert create([0, ndev]) == _capi.ERR_INVALID and not h.value
    assert create([0]) == 0 and h.value
    assert lib.cheetah_ctx_num_devices(h) == 1
    # one context, one device: a product against the oracle-checked KAT (Fp6 mul, src/fp6.rs:1602-1627)
    a = np.array([[7311251207099623982, 11101156575191577169, 3471450285948349450, 18081717484378824777,
                   17271827452459074620, 13245881484175979175]], np.uint64)
    b = np.array([[10376570592177434053, 2266730037751779421, 17604547680137376634, 8522386576290610241,
                   4622828580943454349, 3310945140500303981]], np.uint64)
    c = np.array([[16699204345089849089, 12290068205868725576, 15503892176250492698, 16153850030910369456,
                   4357822170351527454, 15609745236686299732]], np.uint64)
    out = np.zeros_like(a)
    assert lib.cheetah_fp6_mul_batch(h, a.ctypes.data, b.ctypes.data, out.ctypes.data, 1) == 0
    assert (out == c).all()
    assert lib.cheetah_ctx_destroy(h) == 0
    if ndev >= 2:  # a real two-device context where the box has one
        assert create([0, 1]) == 0 and lib.cheetah_ctx_num_devices(h) == 2
        out[:] = 0
        assert lib.cheetah_fp6_mul_batch(h, a.ctypes.data, b.ctypes.data, out.ctypes.data, 1) == 0 and (out == c).all()
        assert lib.cheetah_ctx_destroy(h) == 0
    assert lib.cheetah_debug_ctx_create_replicated(0, 0, C.byref(h)) == _capi.ERR_INVALID


@pytest.mark.parametrize("n", [1, 5, 256, 777, 3000])
def test_sharded_context_equals_single_device_host_arrays(eng, eng3, n):
    """Every host-pointer entry point on a 3-shard context == the single-device context (ragged n: the last
    shard is short or empty)."""
    assert eng3.num_devices == 3
    a, b = synth.fp6_elements(11, n), synth.fp6_elements(12, n)
    a[0] = 0
    assert (eng3.fp6_mul_batch(a, b) == eng.fp6_mul_batch(a, b)).all()
    assert (eng3.fp6_square_batch(a) == eng.fp6_square_batch(a)).all()
    for f in ("fp6_invert_batch", "fp6_sqrt_batch"):
        o3, s3 = getattr(eng3, f)(a)
        o1, s1 = getattr(eng, f)(a)
        assert (o3 == o1).all() and (s3 == s1).all()
    k = synth.scalars(13, n)
    pts = points(eng, 14, n)
    for form in (AFFINE, PROJECTIVE, JACOBIAN):
        g3, i3 = eng3.mul_by_generator_batch(k, form)
        g1, i1 = eng.mul_by_generator_batch(k, form)
        assert (g3 == g1).all() and (i3 == i1).all()
    for in_form, out_form in ((AFFINE, AFFINE), (AFFINE, PROJECTIVE), (AFFINE, JACOBIAN)):
        for vt in (False, True):
            f3 = eng3.multiply_vartime_batch if vt else eng3.multiply_batch
            f1 = eng.multiply_vartime_batch if vt else eng.multiply_batch
            r3, j3 = f3(pts, k, in_form, out_form)
            r1, j1 = f1(pts, k, in_form, out_form)
            assert (r3 == r1).all() and (j3 == j1).all(), (in_form, out_form, vt)
    jac, _ = eng.multiply_batch(pts, k, AFFINE, JACOBIAN)
    x3, y3 = eng3.batch_normalize(jac, JACOBIAN), eng.batch_normalize(jac, JACOBIAN)
    assert (x3[0] == y3[0]).all() and (x3[1] == y3[1]).all()
    xy, inf = y3
    for rec, enc in ((97, "to_uncompressed_batch"), (49, "to_compressed_batch")):
        e3, e1 = getattr(eng3, enc)(xy, inf), getattr(eng, enc)(xy, inf)
        assert (e3 == e1).all()
        dec = eng3.from_uncompressed_batch if rec == 97 else eng3.from_compressed_batch
        dxy, dinf, dsome = dec(e1, True)
        assert (dxy == xy).all() and (dinf == inf).all() and dsome.all()
    assert (eng3.is_torsion_free_batch(pts, AFFINE) == 1).all()
    if n >= 2 and n % 2 == 0:
        for f in ("multiply_many_batch", "multiply_many_vartime_batch"):
            m3, mi3 = getattr(eng3, f)(pts, k, 2, AFFINE, AFFINE)
            m1, mi1 = getattr(eng, f)(pts, k, 2, AFFINE, AFFINE)
            assert (m3 == m1).all() and (mi3 == mi1).all(), f
    # group law and the composed a P + b G path (its temporaries live in each shard's context)
    dbl = eng.double_batch(jac, JACOBIAN)
    assert (eng3.double_batch(jac, JACOBIAN) == dbl).all()
    assert (eng3.add_batch(jac, dbl, JACOBIAN) == eng.add_batch(jac, dbl, JACOBIAN)).all()
    assert (eng3.add_batch(jac, pts, JACOBIAN, AFFINE) == eng.add_batch(jac, pts, JACOBIAN, AFFINE)).all()
    k2 = synth.scalars(15, 2 * n)
    d3, di3 = eng3.multiply_double_with_basepoint_vartime_batch(pts, k2, AFFINE, AFFINE)
    d1, di1 = eng.multiply_double_with_basepoint_vartime_batch(pts, k2, AFFINE, AFFINE)
    assert (d3 == d1).all() and (di3 == di1).all()


@pytest.mark.parametrize("n", [3, 1000, 2049])
def test_sharded_context_equals_single_device_device_arrays(eng, eng3, n):
    """The `_dev` entry points of a multi-shard context: slices travel to the other shards by peer copies and the
    results come back into the caller's device arrays, asynchronously on the caller's stream."""
    import torch
    a, b = dev(synth.fp6_elements(21, n)), dev(synth.fp6_elements(22, n))
    assert torch.equal(eng3.fp6_mul_batch(a, b).view(torch.int64), eng.fp6_mul_batch(a, b).view(torch.int64))
    k = dev(synth.scalars(23, n))
    pts = dev(points(eng, 24, n))
    for form in (AFFINE, PROJECTIVE, JACOBIAN):
        g3, i3 = eng3.mul_by_generator_batch(k, form)
        g1, i1 = eng.mul_by_generator_batch(k, form)
        assert torch.equal(g3.view(torch.int64), g1.view(torch.int64)) and torch.equal(i3, i1)
    for out_form in (AFFINE, PROJECTIVE, JACOBIAN):
        r3, j3 = eng3.multiply_batch(pts, k, AFFINE, out_form)
        r1, j1 = eng.multiply_batch(pts, k, AFFINE, out_form)
        assert torch.equal(r3.view(torch.int64), r1.view(torch.int64)) and torch.equal(j3, j1)
    r3, j3 = eng3.multiply_vartime_batch(pts, k, AFFINE, PROJECTIVE)
    x3, y3 = eng3.batch_normalize(r3, PROJECTIVE), eng.batch_normalize(r3, PROJECTIVE)
    assert torch.equal(x3[0].view(torch.int64), y3[0].view(torch.int64)) and torch.equal(x3[1], y3[1])
    e3, e1 = eng3.to_uncompressed_batch(*y3), eng.to_uncompressed_batch(*y3)
    assert torch.equal(e3, e1)
    s3, s1 = eng3.add_batch(r3, pts, PROJECTIVE, AFFINE), eng.add_batch(r3, pts, PROJECTIVE, AFFINE)
    assert torch.equal(s3.view(torch.int64), s1.view(torch.int64))
    k2 = dev(synth.scalars(25, 2 * n))
    d3, di3 = eng3.multiply_double_with_basepoint_vartime_batch(pts, k2, AFFINE, AFFINE)
    d1, di1 = eng.multiply_double_with_basepoint_vartime_batch(pts, k2, AFFINE, AFFINE)
    assert torch.equal(d3.view(torch.int64), d1.view(torch.int64)) and torch.equal(di3, di1)
    torch.cuda.synchronize()


# ------------------------------------------------------- host-pointer pipeline -----

def test_host_pipeline_many_chunks_equals_device_path(eng):
    """Batches of several pipeline chunks (2^20 units for the elementwise kernels, 2^18 for the ladders) with a
    ragged tail, through the host-pointer entry points, against the `_dev` entry points on the same data."""
    n = (1 << 21) + 1234
    a, b = synth.fp6_elements(31, n), synth.fp6_elements(32, n)
    got = eng.fp6_mul_batch(a, b)
    want = host(eng.fp6_mul_batch(dev(a), dev(b)))
    assert (got == want).all()
    m = 3 * (1 << 18) + 77
    k = host(eng.synth_scalars(33, m))
    pts_d, _ = eng.mul_by_generator_batch(eng.synth_scalars(34, m), AFFINE)
    pts = host(pts_d)
    got, ginf = eng.multiply_batch(pts, k, AFFINE, AFFINE)
    want, winf = eng.multiply_batch(pts_d, dev(k), AFFINE, AFFINE)
    assert (got == host(want)).all() and (ginf == host(winf)).all()
    # a second, smaller call re-uses the staging buffers
    got2, _ = eng.multiply_batch(pts[:1000], k[:1000], AFFINE, AFFINE)
    assert (got2 == got[:1000]).all()
    nxy, ninf = eng.batch_normalize(host(eng.multiply_batch(pts_d[:5000], dev(k[:5000]), AFFINE, JACOBIAN)[0]), JACOBIAN)
    assert (nxy == got[:5000]).all()
    # fixed base from host arrays: the kernels store into the page-locked staging themselves (no device-to-host copy)
    g = 5 * (1 << 18) + 333
    r = host(eng.synth_scalars(35, g))
    for form in (AFFINE, PROJECTIVE, JACOBIAN):
        got, ginf = eng.mul_by_generator_batch(r, form)
        want, winf = eng.mul_by_generator_batch(dev(r), form)
        assert (got == host(want)).all() and (ginf == host(winf)).all(), form
    got2, _ = eng.mul_by_generator_batch(r[:70000], AFFINE)    # one chunk, page-locked; then a small call with direct copies
    got3, _ = eng.mul_by_generator_batch(r[:100], AFFINE)
    want, _ = eng.mul_by_generator_batch(dev(r[:70000]), AFFINE)
    assert (got2 == host(want)).all() and (got3 == host(want)[:100]).all()


# ------------------------------------------------------------ BasePointTable -----

def test_basepoint_table_create_and_multiply_random_bases(eng, ora):
    rng = np.random.default_rng(41)
    bases = host(eng.mul_by_generator_batch(dev(synth.scalars(41, 3)), AFFINE)[0])
    proj_base = host(eng.multiply_batch(dev(bases[:1]), dev(synth.scalars(42, 1)), AFFINE, PROJECTIVE)[0])[0]  # Z != 1
    n = 300
    k = synth.scalars(43, n).copy()
    k[0] = 0
    k[1] = 0
    k[1, 0] = 1
    k[2] = np.frombuffer(((1 << 255) - 1).to_bytes(32, "little"), np.uint8)
    k[3] = np.frombuffer((synth.Q - 1).to_bytes(32, "little"), np.uint8)
    off_aff = bases[2].copy()
    off_aff[0] ^= np.uint64(9)                       # NOT on the curve: both forms must still be the reference comb's values
    off_proj = proj_base.copy()
    off_proj[7] = (int(off_proj[7]) + 3) % P
    assert ora.is_on_curve_batch(off_aff[None, :], 0)[0] == 0 and ora.is_on_curve_batch(off_proj[None, :], 1)[0] == 0
    for base, form in [(bases[0], AFFINE), (bases[1], AFFINE), (proj_base, PROJECTIVE), (IDENT, AFFINE), (off_aff, AFFINE),
                       (off_proj, PROJECTIVE)]:
        t = eng.basepoint_table(base, form)
        wt = ora.basepoint_table_create(base, form)
        assert (t.read() == wt[0]).all()
        assert ((t.read()[:, :6] == 0).all(axis=1) == (wt[1] == 1)).all()   # identities are stored as (0, 1)
        for out_form in (PROJECTIVE, AFFINE):
            want, winf = ora.basepoint_table_multiply_batch(wt, k, out_form)
            got, ginf = t.multiply_batch(k, out_form)                      # host arrays
            assert (got == want).all() and (ginf == winf).all(), (form, out_form)
            gd, gi = t.multiply_batch(dev(k), out_form)                    # device arrays
            assert (host(gd) == want).all() and (host(gi) == winf).all()
        t.close()


def test_basepoint_table_of_generator_is_basepoint_table(eng):
    """BasePointTable::create(g).multiply(s) == BASEPOINT_TABLE.multiply(s) == g.multiply(s) for random s -- the
    reference's cross-path test (src/curve/projective.rs:1328-1351), projective representative included."""
    g = eng.read_table(0)[0]
    t = eng.basepoint_table(g, AFFINE)
    assert (t.read() == eng.read_table(0)).all()
    k = dev(synth.scalars(44, 4096))
    for form in (PROJECTIVE, AFFINE):
        a, ai = t.multiply_batch(k, form)
        b, bi = eng.mul_by_generator_batch(k, form)
        assert (host(a) == host(b)).all() and (host(ai) == host(bi)).all()
    gg = dev(np.repeat(g[None, :], 4096, axis=0))
    c, ci = eng.multiply_batch(gg, k, AFFINE, AFFINE)
    assert (host(c) == host(a)).all()
    t.close()


def test_basepoint_table_on_sharded_context(eng, eng3):
    base = host(eng.mul_by_generator_batch(dev(synth.scalars(45, 1)), AFFINE)[0])[0]
    t1, t3 = eng.basepoint_table(base), eng3.basepoint_table(base)
    k = synth.scalars(46, 1500)
    for form in (PROJECTIVE, AFFINE):
        a, ai = t1.multiply_batch(k, form)
        b, bi = t3.multiply_batch(k, form)
        c, ci = t3.multiply_batch(dev(k), form)
        assert (a == b).all() and (ai == bi).all() and (host(c) == a).all()
    t1.close()
    t3.close()


def test_out_of_range_scalars_are_rejected_the_same_in_both_forms(eng):
    """BASEPOINT_TABLE forms and scalars with the top nibble > 7: the host entry point refuses the
    call; the device entry point rejects the element -- zero limbs, inf = 0xFF -- whatever the output form, and
    needs inf; the neighbours are untouched.  BASEPOINT_LOOKUP (JACOBIAN) takes any scalar."""
    from cheetah_amd import CheetahError
    k = synth.scalars(47, 600).copy()
    bad = [0, 255, 256, 599]
    for i in bad:
        k[i, 31] |= 0x80
    with pytest.raises(CheetahError) as e:
        eng.mul_by_generator_batch(k, AFFINE)
    assert e.value.code == _capi.ERR_SCALAR
    good = k.copy()
    good[:, 31] &= 0x7F
    for form in (AFFINE, PROJECTIVE):
        out, inf = eng.mul_by_generator_batch(dev(k), form)
        ref, rinf = eng.mul_by_generator_batch(dev(good), form)
        out, inf, ref, rinf = host(out), host(inf), host(ref), host(rinf)
        mask = np.zeros(600, bool)
        mask[bad] = True
        assert (inf[mask] == 0xFF).all() and (out[mask] == 0).all()
        assert (out[~mask] == ref[~mask]).all() and (inf[~mask] == rinf[~mask]).all()
    lib = _capi.load()
    kd = dev(k)
    o = dev(np.zeros((600, 18), np.uint64))
    assert lib.cheetah_mul_by_generator_batch_dev(eng._h, kd.data_ptr(), o.data_ptr(), None, 600, PROJECTIVE) == _capi.ERR_INVALID
    assert lib.cheetah_mul_by_generator_batch_dev(eng._h, kd.data_ptr(), o.data_ptr(), None, 600, JACOBIAN) == 0
    eng.synchronize()


# --------------------------------------------------------------- is_on_curve -----

def test_is_on_curve_all_forms(eng, ora):
    n = 200
    k = synth.scalars(51, n)
    aff = points(eng, 52, n)
    proj, _ = eng.multiply_batch(aff, k, AFFINE, PROJECTIVE)
    jac, _ = eng.multiply_batch(aff, k, AFFINE, JACOBIAN)
    for form, pts in ((AFFINE, aff.copy()), (PROJECTIVE, proj.copy()), (JACOBIAN, jac.copy())):
        pts[5, 0] ^= 1                       # off the curve
        pts[6, 7] = (int(pts[6, 7]) + 1) % P
        pts[7, :6] = P                       # non-canonical zero x
        want = ora.is_on_curve_batch(pts, form)
        assert want[5] == 0 and want[6] == 0 and want.sum() >= n - 4
        assert (eng.is_on_curve_batch(pts, form) == want).all()
        assert (host(eng.is_on_curve_batch(dev(pts), form)) == want).all()
    # the point of the curve that is not in the prime-order subgroup (src/curve/projective.rs:1407-1425) is on the curve
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")) as f:
        kats = json.load(f)
    v = kats["non_subgroup_point"]
    pt = np.array([v["x"] + v["y"]], np.uint64)
    assert eng.is_on_curve_batch(pt, AFFINE)[0] == 1 and eng.is_torsion_free_batch(pt, AFFINE)[0] == 0


# ------------------------------------------- low-order inputs (window-table fallback) -----

def test_multiply_low_order_points_fall_back_to_the_complete_formulas(eng, ora, kats):
    """The window table [P .. NP] is built by affine steps with shared inversions; a point of order <= N (or the
    identity) makes a denominator zero and its wave rebuilds the table with the reference's complete additions.
    Points of order 2, 5 and 10 exist on the curve (h = 2 * 5 * 29 * 181 * ...): T_d = (h / d) * (q * N) for the
    reference's point N outside the prime-order subgroup (src/curve/projective.rs:1407-1425).  Mixed into ordinary
    batches (first wave, a later wave), every ladder against the oracle, all forms."""
    h = int.from_bytes(bytes(kats["cofactor_bytes"]), "little") if "cofactor_bytes" in kats else 0x2150b48e071ef610049bc3f5d54304e4a
    v = kats["non_subgroup_point"]
    N = np.array([v["x"] + v["y"]], np.uint64)

    def sc(x):
        return np.frombuffer(int(x).to_bytes(32, "little"), np.uint8).reshape(1, 32).copy()

    # N itself has no 5-torsion component; random points of the curve (AffinePoint::from_x on seeded x) do, 4 times in 5
    xy, xinf, some = ora.affine_from_x_batch(synth.fp6_elements(60, 24))
    cand = np.concatenate([N, xy[(some == 1) & (xinf == 0)]])
    q_cand, _ = ora.multiply_batch(cand, 0, np.repeat(sc(synth.Q), len(cand), axis=0), 0, vartime=True)
    low = []
    for d in (2, 5, 10):
        T, tinf = ora.multiply_batch(q_cand, 0, np.repeat(sc(h // d), len(cand), axis=0), 0, vartime=True)
        chk, cinf = ora.multiply_batch(T, 0, np.repeat(sc(d), len(cand), axis=0), 0, vartime=True)
        assert (cinf == 1).all()                                   # d * T = O for every candidate
        full = [i for i in range(len(cand)) if tinf[i] == 0 and all(
            ora.multiply_batch(T[i:i + 1], 0, sc(e), 0, vartime=True)[1][0] == 0 for e in (1, 2, 5) if e < d and d % e == 0)]
        assert full, d                                             # a point of order exactly d
        low.append(T[full[0]])
    assert (low[0][6:] == 0).all()                                 # order 2: y = 0
    n = 320
    pts = points(eng, 61, n)
    for j, i in enumerate([0, 1, 2, 5, 200, 201, 202, 319]):
        pts[i] = low[j % 3]
    k = synth.scalars(62, n).copy()
    k[1] = 0xFF                                                   # 16-entry table for the first wave
    k[202, 31] |= 0xF0
    for out_form in (AFFINE, JACOBIAN, PROJECTIVE):
        for vt in (False, True):
            want, winf = ora.multiply_batch(pts, 0, k, out_form, vartime=vt)
            f = eng.multiply_vartime_batch if vt else eng.multiply_batch
            got, ginf = f(dev(pts), dev(k), AFFINE, out_form)
            assert (host(got) == want).all() and (host(ginf) == winf).all(), (out_form, vt)
    want, winf = ora.multiply_many_batch(pts, 0, k, 2, 0)
    got, ginf = eng.multiply_many_batch(dev(pts), dev(k), 2, AFFINE, AFFINE)
    assert (host(got) == want).all() and (host(ginf) == winf).all()
    # Jacobian and projective inputs of the same points: (x l^2, y l^3, l) and (x l, y l, l)
    lam = synth.fp6_elements(63, n)
    l2 = eng.fp6_square_batch(lam)
    l3 = eng.fp6_mul_batch(l2, lam)
    jac = np.concatenate([eng.fp6_mul_batch(pts[:, :6].copy(), l2), eng.fp6_mul_batch(pts[:, 6:].copy(), l3), lam], axis=1)
    prj = np.concatenate([eng.fp6_mul_batch(pts[:, :6].copy(), lam), eng.fp6_mul_batch(pts[:, 6:].copy(), lam), lam], axis=1)
    for i in range(3, n, 97):                                       # the identities of `points`
        jac[i, 12:] = 0
        prj[i, 12:] = 0
    for arr, form in ((jac, JACOBIAN), (prj, PROJECTIVE)):
        for out_form in (AFFINE, form):
            for vt in (False, True):          # vartime: NafLookupTable<8> by affine steps, the same fallback
                want, winf = ora.multiply_batch(arr, form, k, out_form, vartime=vt)
                f = eng.multiply_vartime_batch if vt else eng.multiply_batch
                got, ginf = f(dev(arr), dev(k), form, out_form)
                assert (host(got) == want).all() and (host(ginf) == winf).all(), (form, out_form, vt)
    # the uniform-scalar ladders and the Straus NAF ladder over the same points
    hb = np.repeat(sc(h), n, axis=0)
    want, winf = ora.multiply_batch(pts, 0, hb, PROJECTIVE, vartime=True)
    got, ginf = eng.clear_cofactor_batch(dev(pts), AFFINE, PROJECTIVE)
    assert (host(got) == want).all() and (host(ginf) == winf).all()
    want = ora.multiply_batch(pts, 0, np.repeat(sc(synth.Q), n, axis=0), JACOBIAN, vartime=True)[1]
    assert (host(eng.is_torsion_free_batch(dev(pts), AFFINE)) == want).all()   # whatever the reference's incomplete formulas make of them
    want, winf = ora.multiply_many_batch(pts, 0, k, 2, PROJECTIVE, vartime=True)
    got, ginf = eng.multiply_many_vartime_batch(dev(pts), dev(k), 2, AFFINE, PROJECTIVE)
    assert (host(got) == want).all() and (host(ginf) == winf).all()


# ------------------------------------------- off-curve inputs (window-table fallback) -----

def test_off_curve_inputs_match_the_reference_through_every_ladder(eng, ora):
    """Points that are NOT on the curve reach the ladders through from_uncompressed_unchecked / unchecked decoding.
    The reference fills LookupTable<N> with its complete additions (which use 3B), so what it returns for such a
    point is a definite value -- not a multiple on any curve.  The affine chord / tangent table construction would give
    other values, so an off-curve lane sends its wave to the reference's construction (build_lookup_table).  Perturbed
    affine, projective and Jacobian inputs (first wave, a later wave) through multiply, multiply_vartime,
    multiply_many(_vartime) and multiply_double_with_basepoint_vartime against the oracle, all result forms."""
    n = 330
    pts = points(eng, 71, n)
    for i in (1, 64, 65, 200, 329):
        pts[i, 0] ^= np.uint64(1 + i)                 # x off the curve
    pts[2, 7] = (int(pts[2, 7]) + 5) % P              # y off the curve
    pts[130] = np.array(list(range(1, 13)), np.uint64)
    assert ora.is_on_curve_batch(pts, 0)[[1, 2, 64, 65, 130, 200, 329]].sum() == 0
    k = synth.scalars(72, n).copy()
    k[64] = 0xFF                                      # a 16-entry table next to an off-curve lane
    lam = synth.fp6_elements(73, n)
    l2 = eng.fp6_square_batch(lam)
    l3 = eng.fp6_mul_batch(l2, lam)
    jac = np.concatenate([eng.fp6_mul_batch(pts[:, :6].copy(), l2), eng.fp6_mul_batch(pts[:, 6:].copy(), l3), lam], axis=1)
    prj = np.concatenate([eng.fp6_mul_batch(pts[:, :6].copy(), lam), eng.fp6_mul_batch(pts[:, 6:].copy(), lam), lam], axis=1)
    for i in range(3, n, 97):
        jac[i, 12:] = 0
        prj[i, 12:] = 0
    for arr, form in ((pts, AFFINE), (jac, JACOBIAN), (prj, PROJECTIVE)):
        for out_form in sorted({AFFINE, form if form != AFFINE else JACOBIAN, PROJECTIVE if form == AFFINE else form}):
            for vt in (False, True):
                want, winf = ora.multiply_batch(arr, form, k, out_form, vartime=vt)
                f = eng.multiply_vartime_batch if vt else eng.multiply_batch
                got, ginf = f(dev(arr), dev(k), form, out_form)
                assert (host(got) == want).all() and (host(ginf) == winf).all(), (form, out_form, vt)
        m = 2
        for vt in (False, True):
            want, winf = ora.multiply_many_batch(arr, form, k, m, 0, vartime=vt)
            f = eng.multiply_many_vartime_batch if vt else eng.multiply_many_batch
            got, ginf = f(dev(arr), dev(k), m, form, AFFINE)
            assert (host(got) == want).all() and (host(ginf) == winf).all(), (form, "many", vt)
        kk = np.ascontiguousarray(np.stack([k, synth.scalars(74, n)], axis=1).reshape(2 * n, 32))
        want, winf = ora.multiply_many_batch(arr, form, kk, 1, 0, vartime=True, with_basepoint=True)
        got, ginf = eng.multiply_double_with_basepoint_vartime_batch(dev(arr), dev(kk), form, AFFINE)
        assert (host(got) == want).all() and (host(ginf) == winf).all(), (form, "with_basepoint")


def test_point_independent_tables_return_the_same_results(eng, ora):
    """cheetah_ctx_set_point_independent_tables: every wave of the constant-time ladders builds its window tables the
    reference's way (LookupTable::from, src/lookup.rs:25-37) instead of choosing per wave.  Ordinary points, identities,
    a point off the curve and wide scalars in one batch, all forms, single- and multi-scalar: bit-equal to the oracle
    with the switch on, and to the default path."""
    n = 300
    pts = points(eng, 81, n)
    pts[7] = IDENT
    pts[70, 3] ^= np.uint64(6)                          # off the curve
    k = synth.scalars(82, n).copy()
    k[5] = 0xFF
    k[6] = 0
    lam = synth.fp6_elements(83, n)
    prj = np.concatenate([eng.fp6_mul_batch(pts[:, :6].copy(), lam), eng.fp6_mul_batch(pts[:, 6:].copy(), lam), lam], axis=1)
    prj[7, 12:] = 0
    base = {}
    for arr, form, out_form in ((pts, AFFINE, AFFINE), (pts, AFFINE, JACOBIAN), (prj, PROJECTIVE, PROJECTIVE), (prj, PROJECTIVE, AFFINE)):
        base[(form, out_form)] = tuple(host(x) for x in eng.multiply_batch(dev(arr), dev(k), form, out_form))
    base["many"] = tuple(host(x) for x in eng.multiply_many_batch(dev(pts), dev(k), 2, AFFINE, AFFINE))
    eng.set_point_independent_tables(True)
    try:
        for arr, form, out_form in ((pts, AFFINE, AFFINE), (pts, AFFINE, JACOBIAN), (prj, PROJECTIVE, PROJECTIVE), (prj, PROJECTIVE, AFFINE)):
            want, winf = ora.multiply_batch(arr, form, k, out_form, vartime=False)
            got, ginf = (host(x) for x in eng.multiply_batch(dev(arr), dev(k), form, out_form))
            assert (got == want).all() and (ginf == winf).all(), (form, out_form)
            assert (got == base[(form, out_form)][0]).all() and (ginf == base[(form, out_form)][1]).all()
        got, ginf = (host(x) for x in eng.multiply_many_batch(dev(pts), dev(k), 2, AFFINE, AFFINE))
        assert (got == base["many"][0]).all() and (ginf == base["many"][1]).all()
        # the vartime entry points do not look at the switch
        want, winf = ora.multiply_batch(pts, AFFINE, k, AFFINE, vartime=True)
        got, ginf = (host(x) for x in eng.multiply_vartime_batch(dev(pts), dev(k), AFFINE, AFFINE))
        assert (got == want).all() and (ginf == winf).all()
    finally:
        eng.set_point_independent_tables(False)


def _low_order_points(ora, kats):
    """points of order exactly 2, 5 and 10 (the cofactor is 2 * 5 * 29 * ...), from seeded points of the curve"""
    h = 0x2150b48e071ef610049bc3f5d54304e4a

    def sc(x):
        return np.frombuffer(int(x).to_bytes(32, "little"), np.uint8).reshape(1, 32).copy()

    xy, xinf, some = ora.affine_from_x_batch(synth.fp6_elements(60, 24))
    cand = xy[(some == 1) & (xinf == 0)]
    q_cand, _ = ora.multiply_batch(cand, 0, np.repeat(sc(synth.Q), len(cand), axis=0), 0, vartime=True)
    low = {}
    for d in (2, 5, 10):
        T, tinf = ora.multiply_batch(q_cand, 0, np.repeat(sc(h // d), len(cand), axis=0), 0, vartime=True)
        for i in range(len(cand)):
            if tinf[i] == 0 and all(ora.multiply_batch(T[i:i + 1], 0, sc(e), 0, vartime=True)[1][0] == 0 for e in (1, 2, 5) if e < d and d % e == 0):
                low[d] = T[i].copy()
                break
    assert sorted(low) == [2, 5, 10]
    return low


def test_low_order_points_alone_in_their_wave(eng, ora, kats):
    """Quirk Q7 and its neighbours, each ALONE in a wave of ordinary points (so that no other lane's condition sends the
    wave to the reference's table construction): the point of order 2 and a point of order 10, for which the reference's
    tables are not the tables of true multiples (its complete addition meets its exceptional pair), a point of order 5,
    the identity.  Small scalars that use each table entry in turn, every ladder and form, against the oracle.  (Found
    in round 3: the order-10 case had gone through the affine table construction, which builds TRUE multiples.)"""
    low = _low_order_points(ora, kats)
    n = 384
    base = points(eng, 91, n)
    where = {10: IDENT, 70: low[2], 200: low[10], 300: low[5]}
    h = 0x2150b48e071ef610049bc3f5d54304e4a

    def sc(x):
        return np.frombuffer(int(x).to_bytes(32, "little"), np.uint8).copy()

    rng = np.random.default_rng(92)
    for kk in (7, 9, 0x1f, 0x5b, 0x99, int.from_bytes(rng.bytes(31), "little")):
        pts = base.copy()
        k = synth.scalars(93, n).copy()
        for i, pt in where.items():
            pts[i] = pt
            k[i] = sc(kk)
        lam = synth.fp6_elements(94, n)
        l2 = eng.fp6_square_batch(lam)
        l3 = eng.fp6_mul_batch(l2, lam)
        jac = np.concatenate([eng.fp6_mul_batch(pts[:, :6].copy(), l2), eng.fp6_mul_batch(pts[:, 6:].copy(), l3), lam], axis=1)
        prj = np.concatenate([eng.fp6_mul_batch(pts[:, :6].copy(), lam), eng.fp6_mul_batch(pts[:, 6:].copy(), lam), lam], axis=1)
        jac[10, 12:] = 0
        prj[10, 12:] = 0
        for arr, form in ((pts, AFFINE), (jac, JACOBIAN), (prj, PROJECTIVE)):
            for out_form in sorted({AFFINE, JACOBIAN if form != PROJECTIVE else PROJECTIVE, PROJECTIVE if form != JACOBIAN else JACOBIAN}):
                for vt in (False, True):
                    want, winf = ora.multiply_batch(arr, form, k, out_form, vartime=vt)
                    f = eng.multiply_vartime_batch if vt else eng.multiply_batch
                    got, ginf = f(dev(arr), dev(k), form, out_form)
                    bad = np.nonzero((host(got) != want).any(axis=1) | (host(ginf) != winf))[0]
                    assert len(bad) == 0, (hex(kk), form, out_form, vt, bad[:6])
            for out_form in (AFFINE, PROJECTIVE if form != JACOBIAN else JACOBIAN):
                for vt in (False, True):
                    want, winf = ora.multiply_many_batch(arr, form, k, 2, out_form, vartime=vt)
                    f = eng.multiply_many_vartime_batch if vt else eng.multiply_many_batch
                    got, ginf = f(dev(arr), dev(k), 2, form, out_form)
                    bad = np.nonzero((host(got) != want).any(axis=1) | (host(ginf) != winf))[0]
                    assert len(bad) == 0, (hex(kk), form, out_form, vt, "many", bad[:6])
            kk2 = np.ascontiguousarray(np.stack([k, synth.scalars(95, n)], axis=1).reshape(2 * n, 32))
            for out_form in (AFFINE, PROJECTIVE if form != JACOBIAN else JACOBIAN):
                want, winf = ora.multiply_many_batch(arr, form, kk2, 1, out_form, vartime=True, with_basepoint=True)
                got, ginf = eng.multiply_double_with_basepoint_vartime_batch(dev(arr), dev(kk2), form, out_form)
                bad = np.nonzero((host(got) != want).any(axis=1) | (host(ginf) != winf))[0]
                assert len(bad) == 0, (hex(kk), form, out_form, "a P + b G", bad[:6])
    # the uniform-scalar ladders
    hb = np.repeat(sc(h).reshape(1, 32), n, axis=0)
    for out_form in (AFFINE, PROJECTIVE):
        want, winf = ora.multiply_batch(pts, 0, hb, out_form, vartime=True)
        got, ginf = eng.clear_cofactor_batch(dev(pts), AFFINE, out_form)
        assert (host(got) == want).all() and (host(ginf) == winf).all(), out_form
    want = ora.multiply_batch(pts, 0, np.repeat(sc(synth.Q).reshape(1, 32), n, axis=0), JACOBIAN, vartime=True)[1]
    assert (host(eng.is_torsion_free_batch(dev(pts), AFFINE)) == want).all()
    assert want[70] == 1                      # quirk Q7: the reference calls the point of order 2 torsion free
    # BasePointTable over such a base: the reference's comb, whatever the result form
    ks = synth.scalars(96, 200).copy()
    ks[:, 31] &= 0x7F
    for i in range(40):
        ks[i] = sc(i)
    for d in (2, 5, 10):
        tab = ora.basepoint_table_create(low[d], AFFINE)
        t = eng.basepoint_table(low[d].reshape(1, 12), AFFINE)
        try:
            for of in (AFFINE, PROJECTIVE):
                want = ora.basepoint_table_multiply_batch(tab, ks, of)
                got = t.multiply_batch(dev(ks), of)
                assert (host(got[0]) == want[0]).all() and (host(got[1]) == want[1]).all(), (d, of)
        finally:
            t.close()

