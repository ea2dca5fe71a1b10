#!/usr/bin/env python3
"""Randomised differential run, GPU against the oracle: random entry point, forms, batch size, with special points
(identities in every encoding, G, -G, repeated points, points that are NOT on the curve) and special scalars (0, 1,
q - 1, q, 2^255 - 1, bit 255 set) mixed into seeded batches.
usage: tests/fuzz_gpu.py <seconds> [seed] [max batch size, default 1500]: prints one line per mismatch and a summary;
exit 1 on any mismatch.  tests/test_gpu_fuzz.py runs fuzz() for a fixed number of runs per seed (deterministic) in the
GPU suite; the open-ended form is for soak runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

P = 0xFFFFFFFF00000001
H_COFACTOR = 0x2150B48E071EF610049BC3F5D54304E4A


def fuzz(budget=60.0, seed=1, max_n=1500, max_runs=None, engine=None):
    """runs until `budget` seconds or `max_runs` cases have passed; returns (runs, [finding lines])"""
    import torch
    from cheetah_amd import AFFINE, JACOBIAN, PROJECTIVE, Engine, synth
    from oracle import binding as ora
    Q = synth.Q
    rng = np.random.default_rng(seed)
    ora.build()
    eng = Engine(0) if engine is None else engine
    G = ora.table("ora_generator_affine", 1)[0]
    IDENT = np.array([0] * 6 + [1] + [0] * 5, np.uint64)


    def dev(a):
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.int64)).cuda().view(torch.uint64) if a.dtype == np.uint64 else torch.from_numpy(a).cuda()


    def host(t):
        return t.view(torch.int64).cpu().numpy().view(np.uint64) if t.dtype == torch.uint64 else t.cpu().numpy()


    def le(k):
        return np.frombuffer(int(k).to_bytes(32, "little"), np.uint8)


    def scalars(n):
        sc = synth.scalars(int(rng.integers(1, 1 << 30)), n).copy()
        special = [0, 1, 2, 3, 7, 9, 15, 16, 0x5b, 0x99, Q - 1, Q, (1 << 255) - 1, (1 << 255) + 5, (1 << 256) - 1, 1 << 255]
        for _ in range(int(rng.integers(0, 6))):
            k = special[int(rng.integers(0, len(special)))]
            if k >= 1 << 255 and rng.random() < 0.5:
                continue
            sc[int(rng.integers(0, n))] = le(k)
        return sc


    def _low_order():
        """points of order 2, 5 and 10 (quirk Q7: the reference's tables of the even ones are not tables of multiples)"""
        sc = lambda x: np.frombuffer(int(x).to_bytes(32, "little"), np.uint8).reshape(1, 32).copy()
        xy, xinf, some = ora.affine_from_x_batch(synth.fp6_elements(60, 24))
        cand = xy[(some == 1) & (xinf == 0)]
        qc, _ = ora.multiply_batch(cand, 0, np.repeat(sc(Q), len(cand), axis=0), 0, vartime=True)
        out = []
        for d in (2, 5, 10):
            T, tinf = ora.multiply_batch(qc, 0, np.repeat(sc(H_COFACTOR // d), len(cand), axis=0), 0, vartime=True)
            out += [T[i].copy() for i in range(len(cand)) if tinf[i] == 0][:2]
        return out

    low_order = _low_order()

    def affine_points(n, off_curve=False):
        aff, _ = ora.mul_by_generator_batch(synth.scalars(int(rng.integers(1, 1 << 30)), n), 0)
        aff = aff.copy()
        if rng.random() < 0.35:                                   # a point of small order, often alone in its wave
            aff[int(rng.integers(0, n))] = low_order[int(rng.integers(0, len(low_order)))]
        for _ in range(int(rng.integers(0, 6))):
            i = int(rng.integers(0, n))
            c = int(rng.integers(0, 5 if off_curve else 4))
            if c == 4:                                            # one coordinate limb moved: off the curve
                l = int(rng.integers(0, 12))
                aff[i, l] = (int(aff[i, l]) + int(rng.integers(1, 1 << 40))) % P
                continue
            if c == 0:
                aff[i] = IDENT
            elif c == 1:
                aff[i] = G
            elif c == 2:
                aff[i] = np.concatenate([G[:6], (np.uint64(P) - G[6:]) % np.uint64(P)])
            else:
                aff[i] = aff[int(rng.integers(0, n))]
        return aff


    def to_form(aff, form):
        if form == AFFINE:
            return aff
        n = len(aff)
        z = synth.fp6_elements(int(rng.integers(1, 1 << 30)), n).reshape(n, 6)
        x, y = np.ascontiguousarray(aff[:, :6]), np.ascontiguousarray(aff[:, 6:])
        if form == PROJECTIVE:
            out = np.concatenate([ora.fp6_mul_batch(x, z).reshape(n, 6), ora.fp6_mul_batch(y, z).reshape(n, 6), z], axis=1)
        else:
            z2 = ora.fp6_square_batch(z).reshape(n, 6)
            out = np.concatenate([ora.fp6_mul_batch(x, z2).reshape(n, 6), ora.fp6_mul_batch(y, ora.fp6_mul_batch(z2, z).reshape(n, 6)).reshape(n, 6), z], axis=1)
        out = np.ascontiguousarray(out)
        ident = (aff[:, :6] == 0).all(axis=1) & (aff[:, 6] == 1) & (aff[:, 7:] == 0).all(axis=1)
        for i in np.nonzero(ident)[0]:
            c = int(rng.integers(0, 3))
            if c == 0:
                out[i, 12:] = 0                                   # (X, Y, 0), arbitrary X, Y
            elif c == 1:
                out[i] = [0] * 6 + [1] + [0] * 5 + [0] * 6 if form == PROJECTIVE else [1] + [0] * 5 + [1] + [0] * 5 + [0] * 6
            else:
                out[i, 12:] = P                                   # z = 0 written as p
        return out


    def pick_forms():
        in_form = int(rng.integers(0, 3))
        outs = [AFFINE] + ([PROJECTIVE] if in_form != JACOBIAN else []) + ([JACOBIAN] if in_form != PROJECTIVE else [])
        return in_form, outs[int(rng.integers(0, len(outs)))]


    bad = runs = 0
    findings = []
    t_start = time.time()
    t_end = t_start + budget
    t_note = time.time() + 30
    while time.time() < t_end and (max_runs is None or runs < max_runs):
        if time.time() > t_note:
            print("... %d runs, %d findings" % (runs, bad), flush=True)
            t_note = time.time() + 30
        n = int(rng.integers(1, max_n))
        case = int(rng.integers(0, 21))
        in_form, out_form = pick_forms()
        # points off the curve go where the reference computes a definite value for them too: the ladders, the group
        # law, the predicates (not the encodings, whose input is a valid point by contract)
        aff = affine_points(n, off_curve=case in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 14, 16) and rng.random() < 0.5)
        pts = to_form(aff, in_form)
        use_host = rng.random() < 0.25
        D = (lambda a: a) if use_host else dev
        H = (lambda a: a) if use_host else host
        name = "?"
        try:
            if case in (0, 1):
                vt = case == 1
                k = scalars(n)
                name = "multiply%s %d->%d" % ("_vartime" if vt else "", in_form, out_form)
                want = ora.multiply_batch(pts, in_form, k, out_form, vartime=vt)
                got = (eng.multiply_vartime_batch if vt else eng.multiply_batch)(D(pts), D(k), in_form, out_form)
            elif case in (2, 3):
                vt = case == 3
                m = int(rng.integers(1, 4))
                if n < m:
                    continue
                g = n // m
                k = scalars(g * m)
                p = pts[: g * m]
                name = "multiply_many%s m=%d %d->%d" % ("_vartime" if vt else "", m, in_form, out_form)
                want = ora.multiply_many_batch(p, in_form, k, m, out_form, vartime=vt)
                got = (eng.multiply_many_vartime_batch if vt else eng.multiply_many_batch)(D(p), D(k), m, in_form, out_form)
            elif case == 4:
                k = scalars(2 * n)
                name = "a P + b G %d->%d" % (in_form, out_form)
                want = ora.multiply_many_batch(pts, in_form, k, 1, out_form, vartime=True, with_basepoint=True)
                got = eng.multiply_double_with_basepoint_vartime_batch(D(pts), D(k), in_form, out_form)
            elif case == 5:
                form = PROJECTIVE if in_form == AFFINE else in_form
                p = to_form(aff, form)
                name = "batch_normalize %d" % form
                want = ora.batch_normalize(p, form == JACOBIAN)
                got = eng.batch_normalize(D(p), form)
            elif case == 6:
                form = PROJECTIVE if in_form == AFFINE else in_form
                p, q = to_form(aff, form), to_form(affine_points(n), form)
                if rng.random() < 0.5:
                    name = "add %d" % form
                    want = (ora.add_batch(p, form, q, form),)
                    got = (eng.add_batch(D(p), D(q), form),)
                else:
                    qa = affine_points(n)
                    name = "add_mixed %d" % form
                    want = (ora.add_batch(p, form, qa, AFFINE),)
                    got = (eng.add_batch(D(p), D(qa), form, AFFINE),)
            elif case == 8:
                form = int(rng.integers(0, 3))
                p = to_form(aff, form)
                other = affine_points(n)
                mix = np.where((rng.random(n) < 0.5)[:, None], aff, other)
                q = to_form(np.ascontiguousarray(mix), form)
                name = "eq %d" % form
                want = (ora.eq_batch(p, q, form),)
                got = (eng.eq_batch(D(p), D(q), form),)
            elif case == 9:
                form = PROJECTIVE if in_form == AFFINE else in_form
                p = to_form(aff, form)
                kk = int(rng.integers(1, 5))
                name = "double_multi(%d) %d" % (kk, form)
                want = (ora.double_multi_batch(p, form, kk), ora.neg_batch(p, form))
                got = (eng.double_multi_batch(D(p), kk, form), eng.neg_batch(D(p), form))
            elif case == 10:
                hb = np.tile(le(H_COFACTOR), (n, 1))
                qb = np.tile(le(Q), (n, 1))
                name = "clear_cofactor / is_torsion_free %d" % in_form
                w1 = ora.multiply_batch(pts, in_form, hb, AFFINE, vartime=True)
                w2 = ora.multiply_batch(pts, in_form, qb, PROJECTIVE if in_form == PROJECTIVE else JACOBIAN, vartime=True)[1]
                g1 = eng.clear_cofactor_batch(D(pts), in_form, AFFINE)
                want = (w1[0], w1[1], w2)
                got = (g1[0], g1[1], eng.is_torsion_free_batch(D(pts), in_form))
            elif case == 11:
                inf = ((aff[:, :6] == 0).all(axis=1) & (aff[:, 6] == 1) & (aff[:, 7:] == 0).all(axis=1)).astype(np.uint8)
                comp = bool(rng.integers(0, 2))
                name = "encode / decode %s" % ("compressed" if comp else "uncompressed")
                enc = ora.encode_batch(aff, inf, comp)
                e = (eng.to_compressed_batch if comp else eng.to_uncompressed_batch)(D(aff), D(inf))
                bad_rec = enc.copy()
                bad_rec[:: 7, int(rng.integers(0, enc.shape[1]))] ^= 1 << int(rng.integers(0, 8))   # most of these are rejected
                chk = bool(rng.integers(0, 2))
                w = (ora.decode_compressed_batch if comp else ora.decode_uncompressed_batch)(bad_rec, chk)
                d = (eng.from_compressed_batch if comp else eng.from_uncompressed_batch)(D(bad_rec), chk)
                want = (enc,) + tuple(w)
                got = (e,) + tuple(d)
            elif case == 12:
                a = synth.fp6_elements(int(rng.integers(1, 1 << 30)), n).reshape(n, 6).copy()
                b = synth.fp6_elements(int(rng.integers(1, 1 << 30)), n).reshape(n, 6).copy()
                edge = np.array([0, 1, P - 1, P, 2**64 - 1, 2**32, 2**32 - 1, P + 1], np.uint64)
                for _ in range(int(rng.integers(0, 8))):
                    (a if rng.random() < 0.5 else b)[int(rng.integers(0, n)), int(rng.integers(0, 6))] = edge[int(rng.integers(0, len(edge)))]
                if rng.random() < 0.3:
                    a[int(rng.integers(0, n))] = 0                      # a zero: invert -> none
                name = "fp6 mul / square / invert"
                wi = ora.fp6_invert_batch(a)
                gi = eng.fp6_invert_batch(D(a))
                want = (ora.fp6_mul_batch(a, b).reshape(n, 6), ora.fp6_square_batch(a).reshape(n, 6), wi[0].reshape(n, 6), wi[1])
                got = (eng.fp6_mul_batch(D(a), D(b)), eng.fp6_square_batch(D(a)), gi[0], gi[1])
            elif case == 13:
                m_ = min(n, 400)
                a = rng.integers(0, 2**64, m_, dtype=np.uint64)
                b = rng.integers(0, 2**64, m_, dtype=np.uint64)
                edge = [0, 1, P - 1, P, P + 1, 2**64 - 1, 2**32, 2**32 - 1]
                for _ in range(int(rng.integers(0, 8))):
                    (a if rng.random() < 0.5 else b)[int(rng.integers(0, m_))] = edge[int(rng.integers(0, len(edge)))]
                op = ["add", "sub", "neg", "double", "triple", "mul", "square", "mul_by_u32", "invert", "canonical"][int(rng.integers(0, 10))]
                sc = int(rng.integers(0, 2**32))
                ai, bi = [int(v) for v in a], [int(v) for v in b]
                f = {"add": lambda x, y: x + y, "sub": lambda x, y: x - y, "neg": lambda x, y: -x, "double": lambda x, y: 2 * x,
                     "triple": lambda x, y: 3 * x, "mul": lambda x, y: x * y, "square": lambda x, y: x * x,
                     "mul_by_u32": lambda x, y: x * sc, "invert": lambda x, y: pow(x, P - 2, P), "canonical": lambda x, y: x}[op]
                name = "fp_op %s" % op
                w = np.array([f(x, y) % P for x, y in zip(ai, bi)], np.uint64)
                g = eng.fp_op_batch(op, D(a), D(b) if op in ("add", "sub", "mul") else None, sc)
                if op == "invert":
                    want = (w, np.array([1 if x % P else 0 for x in ai], np.uint8))
                    got = g
                else:
                    want, got = (w,), (g,)
            elif case == 14:
                p = pts.copy()
                for _ in range(int(rng.integers(0, 6))):
                    p[int(rng.integers(0, n)), int(rng.integers(0, p.shape[1]))] ^= np.uint64(1 << int(rng.integers(0, 63)))
                name = "is_on_curve %d" % in_form
                want = (ora.is_on_curve_batch(p, in_form),)
                got = (eng.is_on_curve_batch(D(p), in_form),)
            elif case == 15:
                bf = AFFINE if in_form == JACOBIAN else in_form
                base = to_form(affine_points(1), bf)
                k = scalars(n)
                k[:, 31] &= 0x7F
                of = [AFFINE, PROJECTIVE][int(rng.integers(0, 2))]
                name = "BasePointTable::create(P).multiply %d->%d" % (bf, of)
                tab = ora.basepoint_table_create(base[0], bf)
                t = eng.basepoint_table(base, bf)
                try:
                    want = (tab[0],) + tuple(ora.basepoint_table_multiply_batch(tab, k, of))
                    got = (t.read(),) + tuple(H(x) for x in t.multiply_batch(D(k), of))
                    got = tuple(np.asarray(x) for x in got)
                    H = lambda a: a
                finally:
                    t.close()
            elif case == 16:
                form = PROJECTIVE if in_form == AFFINE else in_form
                p = to_form(aff, form)
                qa = affine_points(n)
                kk = int(rng.integers(1, 5))
                name = "unchecked double_multi(%d) / add_mixed %d" % (kk, form)
                want = (ora.double_multi_unchecked_batch(p, form, kk), ora.add_mixed_unchecked_batch(p, form, qa))
                got = (eng.double_multi_unchecked_batch(D(p), kk, form), eng.add_mixed_unchecked_batch(D(p), D(qa), form))
            elif case == 17:
                m_ = min(n, 600)
                a = rng.integers(0, 2**64, m_, dtype=np.uint64)
                a[: m_ // 3] = [pow(int(v), 2, P) for v in a[: m_ // 3]]       # squares among them
                for _ in range(int(rng.integers(0, 5))):
                    a[int(rng.integers(0, m_))] = [0, 1, 7, P, P - 1, 2**64 - 1][int(rng.integers(0, 6))]
                e = int(rng.integers(0, 2**64, dtype=np.uint64)) >> int(rng.integers(0, 64))
                name = "fp sqrt / exp"
                want = tuple(ora.fp_sqrt(a)) + (np.array([pow(int(v), e, P) for v in a], np.uint64),)
                got = tuple(eng.fp_op_batch("sqrt", D(a))) + (eng.fp_op_batch("exp", D(a), scalar=e),)
            elif case == 18:
                m_ = min(n, 300)
                a = synth.fp6_elements(int(rng.integers(1, 1 << 30)), m_).reshape(m_, 6).copy()
                if rng.random() < 0.3:
                    a[int(rng.integers(0, m_))] = 0
                power = int.from_bytes(rng.bytes(48), "little") >> int(rng.integers(0, 384))
                name = "fp6 exp"
                want = (ora.fp6_exp_batch(a, power).reshape(m_, 6),)
                got = (eng.fp6_exp_batch(D(a), power),)
            elif case == 19:
                # random_points = the candidate construction, then clear_cofactor: checked here through its two
                # invariants and against clear_cofactor by the oracle on the device's own candidates' images
                m_ = min(n, 500)
                sd, st = int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 40))
                of = [AFFINE, PROJECTIVE, JACOBIAN][int(rng.integers(0, 3))]
                name = "random_points ->%d" % of
                xy, inf = eng.random_points(sd, m_, st, AFFINE)
                other, oinf = eng.random_points(sd, m_, st, of)
                back = eng.batch_normalize(other, of) if of != AFFINE else (other, oinf)
                want = (np.ones(m_, np.uint8), np.ones(m_, np.uint8), host(xy), np.zeros(m_, np.uint8))
                got = (host(eng.is_on_curve_batch(xy, AFFINE)), host(eng.is_torsion_free_batch(xy, AFFINE)), host(back[0]), host(inf))
                H = lambda a: a
            elif case == 20:
                name = "batch_normalize %d with inputs off the curve, z = 0 among them" % in_form
                form = PROJECTIVE if in_form == AFFINE else in_form
                p = to_form(aff, form).copy()
                for _ in range(int(rng.integers(0, 4))):
                    p[int(rng.integers(0, n)), 12:] = 0
                want = ora.batch_normalize(p, form == JACOBIAN)
                got = eng.batch_normalize(D(p), form)
            else:
                k = scalars(n)
                of = [AFFINE, PROJECTIVE, JACOBIAN][int(rng.integers(0, 3))]
                if of != JACOBIAN:
                    k[:, 31] &= 0x7F                              # BasePointTable's range (top nibble <= 7)
                name = "mul_by_generator ->%d" % of
                want = ora.mul_by_generator_batch(k, of)
                got = eng.mul_by_generator_batch(D(k), of)
            runs += 1
            for w, g_ in zip(want, got):
                g_ = H(g_)
                if not np.array_equal(np.asarray(g_).reshape(np.asarray(w).shape), w):
                    idx = np.nonzero((np.asarray(g_).reshape(np.asarray(w).shape) != w).reshape(len(w), -1).any(axis=1))[0]
                    findings.append("MISMATCH %s n=%d host=%s first at %s (seed %d, run %d)" % (name, n, use_host, idx[:5], seed, runs))
                    print(findings[-1], flush=True)
                    bad += 1
                    break
        except Exception as e:  # an error code where the oracle has an answer is a finding too
            findings.append("ERROR %s n=%d host=%s: %s" % (name, n, use_host, e))
            print(findings[-1], flush=True)
            bad += 1
    print("fuzz: %d runs, %d findings, seed %d, %.0f s, batches up to %d" % (runs, bad, seed, time.time() - t_start, max_n), flush=True)
    if engine is None:
        eng.close()
    return runs, findings


if __name__ == "__main__":
    _, found = fuzz(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                    int(sys.argv[3]) if len(sys.argv) > 3 else 1500)
    sys.exit(1 if found else 0)
