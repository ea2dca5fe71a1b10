// Fp6 = Fp[u]/(u^6 - 7) device arithmetic (reference: src/fp6.rs).
//
// The reference multiplies with four cubic Karatsuba blocks and reduces every
// one of its 24 Fp products separately (src/fp6.rs:651-834).  On gfx950 a
// multiply costs no more than an add-with-carry (see fp.hip.h), so here the
// product is the plain 36-term schoolbook convolution with the wrap u^6 = 7
// folded into pre-scaled operands (7 * b_j), each output coefficient summed
// unreduced in one Acc3 and reduced once: 6 reductions instead of 24, no
// Karatsuba pre/post additions.  Results are congruent mod p to the
// reference's, coefficient by coefficient.
#pragma once
#include "fp.hip.h"

// Fp6 mul / square are 410 / 305 VALU instructions each.  Inlined into every point
// formula the ladder loop would exceed the instruction cache, so by default they
// are real functions (fp6_mul / fp6_square -> fp6_mul_fn / fp6_square_fn).  Their
// arguments and result cross the call as 6 x u64 vectors: hipcc passes those in
// VGPRs, whereas a second struct argument goes through the stack (scratch memory).
// fp6_mul_inl / fp6_square_inl are the same products inlined at the call site: used
// where one loop body is executed often enough to pay for its size (the run of
// doublings that opens a ladder window, curve.hip.h).
#ifndef CH_FP6_FN
#define CH_FP6_FN __device__ __noinline__
#endif

namespace cheetah {

struct Fp6 {
    u64 c[6];  // c0..c5, low degree first (src/fp6.rs:49-62)
};

CH_DEV Fp6 fp6_zero() {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = 0;
    return r;
}
CH_DEV Fp6 fp6_one() {
    Fp6 r = fp6_zero();
    r.c[0] = 1;
    return r;
}

#define CH_FP6_COEFFWISE2(NAME, OP)                      \
    CH_DEV Fp6 NAME(const Fp6& a, const Fp6& b) {        \
        Fp6 r;                                           \
        _Pragma("unroll") for (int i = 0; i < 6; i++)    \
            r.c[i] = OP(a.c[i], b.c[i]);                 \
        return r;                                        \
    }
#define CH_FP6_COEFFWISE1(NAME, OP)                      \
    CH_DEV Fp6 NAME(const Fp6& a) {                      \
        Fp6 r;                                           \
        _Pragma("unroll") for (int i = 0; i < 6; i++)    \
            r.c[i] = OP(a.c[i]);                         \
        return r;                                        \
    }
CH_DEV Fp6 fp6_add(const Fp6& a, const Fp6& b) {  // src/fp6.rs:994-1003
    Fp6 r = a;
    fp_add3(r.c[0], r.c[1], r.c[2], b.c[0], b.c[1], b.c[2]);
    fp_add3(r.c[3], r.c[4], r.c[5], b.c[3], b.c[4], b.c[5]);
    return r;
}
CH_DEV Fp6 fp6_sub(const Fp6& a, const Fp6& b) {  // src/fp6.rs:1007-1016
    Fp6 r = a;
    fp_sub3(r.c[0], r.c[1], r.c[2], b.c[0], b.c[1], b.c[2]);
    fp_sub3(r.c[3], r.c[4], r.c[5], b.c[3], b.c[4], b.c[5]);
    return r;
}
CH_FP6_COEFFWISE1(fp6_neg, fp_neg)          // src/fp6.rs:1020-1029
CH_DEV Fp6 fp6_double(const Fp6& a) { return fp6_add(a, a); }  // src/fp6.rs:968-977
CH_FP6_COEFFWISE1(fp6_halve, fp_halve)      // a / 2 (no counterpart: see mjac_double_unchecked)
CH_FP6_COEFFWISE1(fp6_triple, fp_triple)    // src/fp6.rs:981-990
CH_FP6_COEFFWISE1(fp6_canonical, fp_canonical)  // src/fp6.rs:373-382

CH_DEV Fp6 fp6_mul_small(const Fp6& a, u32 k) {  // src/fp6.rs:428-444 (k < 2^31)
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul_small(a.c[i], k);
    return r;
}
CH_DEV Fp6 fp6_mul_by_fp(const Fp6& a, u64 k) {  // src/fp6.rs:448-464
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul(a.c[i], k);
    return r;
}
// Branch-free on purpose (one OR over the canonical limbs): written as a chain of `&&` the compiler turned it into a
// chain of divergent branches, i.e. control flow on the operand (tools/check_ct_isa.py).
CH_DEV bool fp6_is_zero(const Fp6& a) {  // src/fp6.rs:362-369
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) acc |= fp_nonzero_bits(a.c[i]);
    return acc == 0;
}
// subtle's conditional_select(a, b, choice) = choice ? b : a   (src/fp6.rs:247-258)
CH_DEV Fp6 fp6_select(const Fp6& a, const Fp6& b, bool choice) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = choice ? b.c[i] : a.c[i];
    return r;
}

// a * b in Fp6.                                     (src/fp6.rs:651-834)
//   c_k = sum_{i+j=k} a_i b_j + sum_{i+j=k+6} a_i (7 b_j)
typedef u64 fp6v __attribute__((ext_vector_type(6)));
CH_DEV fp6v fp6_to_v(const Fp6& a) {
    fp6v v;
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] = a.c[i];
    return v;
}
CH_DEV Fp6 fp6_from_v(fp6v v) {
    Fp6 a;
#pragma unroll
    for (int i = 0; i < 6; i++) a.c[i] = v[i];
    return a;
}

CH_DEV Fp6 fp6_mul_inl(const Fp6& a, const Fp6& b) {  // the product itself, inlined where it is called
    u64 b7[6];
    b7[0] = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) b7[j] = fp_mul_small(b.c[j], 7u);
    Fp6 r;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        Acc3 acc;
        acc.init2(a.c[0], b.c[k], a.c[1], k >= 1 ? b.c[k - 1] : b7[5]);
#pragma unroll
        for (int i = 2; i < 6; i++) {
            if (i <= k)
                acc.mac(a.c[i], b.c[k - i]);
            else
                acc.mac(a.c[i], b7[k + 6 - i]);
        }
        r.c[k] = acc.reduce();
    }
    return r;
}
CH_FP6_FN fp6v fp6_mul_fn(fp6v av, fp6v bv) { return fp6_to_v(fp6_mul_inl(fp6_from_v(av), fp6_from_v(bv))); }
CH_DEV Fp6 fp6_mul(const Fp6& a, const Fp6& b) { return fp6_from_v(fp6_mul_fn(fp6_to_v(a), fp6_to_v(b))); }

// a^2 in Fp6.                                        (src/fp6.rs:838-920)
// 21 products: cross terms use the doubled operand d_i = 2 a_i, wrapped terms
// the scaled operand s_j = 7 a_j.
CH_DEV Fp6 fp6_square_inl(const Fp6& a) {
    u64 d[6], s[6];
#pragma unroll
    for (int i = 0; i < 5; i++) d[i] = fp_double(a.c[i]);
    d[5] = 0;
    s[0] = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) s[j] = fp_mul_small(a.c[j], 7u);
    Fp6 r;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        Acc3 acc;
        int seen = 0;  // every coefficient has at least three terms
        u64 x0 = 0, y0 = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int j = i; j < 6; j++) {
                const bool low = (i + j == k), high = (i + j == k + 6);
                if (low || high) {
                    const u64 x = (i == j) ? a.c[i] : d[i];
                    const u64 y = high ? s[j] : a.c[j];
                    if (seen == 0) {
                        x0 = x;
                        y0 = y;
                    } else if (seen == 1) {
                        acc.init2(x0, y0, x, y);
                    } else {
                        acc.mac(x, y);
                    }
                    seen++;
                }
            }
        }
        r.c[k] = acc.reduce();
    }
    return r;
}
CH_FP6_FN fp6v fp6_square_fn(fp6v av) { return fp6_to_v(fp6_square_inl(fp6_from_v(av))); }
CH_DEV Fp6 fp6_square(const Fp6& a) { return fp6_from_v(fp6_square_fn(fp6_to_v(a))); }

// Frobenius maps: per-coefficient constants (src/fp6.rs:1088-1123).
CH_DEV Fp6 fp6_frobenius(const Fp6& a) {
    Fp6 r;
    r.c[0] = a.c[0];
    r.c[1] = fp_mul(18446744065119617026ull, a.c[1]);
    r.c[2] = fp_mul(18446744065119617025ull, a.c[2]);
    r.c[3] = fp_neg(a.c[3]);
    r.c[4] = fp_mul(4294967295ull, a.c[4]);
    r.c[5] = fp_mul(4294967296ull, a.c[5]);
    return r;
}
CH_DEV Fp6 fp6_frobenius_double(const Fp6& a) {
    Fp6 r;
    r.c[0] = a.c[0];
    r.c[1] = fp_mul(18446744065119617025ull, a.c[1]);
    r.c[2] = fp_mul(4294967295ull, a.c[2]);
    r.c[3] = a.c[3];
    r.c[4] = fp_mul(18446744065119617025ull, a.c[4]);
    r.c[5] = fp_mul(4294967295ull, a.c[5]);
    return r;
}
CH_DEV Fp6 fp6_frobenius_triple(const Fp6& a) {
    Fp6 r;
    r.c[0] = a.c[0];
    r.c[1] = fp_neg(a.c[1]);
    r.c[2] = a.c[2];
    r.c[3] = fp_neg(a.c[3]);
    r.c[4] = a.c[4];
    r.c[5] = fp_neg(a.c[5]);
    return r;
}

// Only coefficient 0 of a * b (the Fp-valued norm step of the inversion).
CH_DEV u64 fp6_mul_c0(const Fp6& a, const Fp6& b) {
    Acc3 acc;
    acc.init2(a.c[0], b.c[0], a.c[1], fp_mul_small(b.c[5], 7u));
#pragma unroll
    for (int i = 2; i < 6; i++) acc.mac(a.c[i], fp_mul_small(b.c[6 - i], 7u));
    return acc.reduce();
}

// a^-1 by the norm to Fp (Itoh-Tsujii), as src/fp6.rs:1035-1049; 0 -> 0 and
// *is_some = false (the reference returns CtOption::none).
CH_DEV Fp6 fp6_invert(const Fp6& a, bool* is_some) {
    Fp6 t0 = fp6_frobenius(a);
    Fp6 t1 = fp6_frobenius_triple(a);
    Fp6 t2 = fp6_mul(t0, t1);
    Fp6 t3 = fp6_frobenius(t2);
    Fp6 t4 = fp6_frobenius_double(t2);
    Fp6 t5 = fp6_mul(t4, t3);
    t5 = fp6_mul(t5, t0);
    u64 n = fp6_mul_c0(t5, a);
    *is_some = !fp_is_zero(n);
    return fp6_mul_by_fp(t5, fp_invert(n));
}


// ---------------------------------------------------------------------------
// Fp3 = Fp[X]/(X^3 - 7), the subfield embedded as (c0, c2, c4) (src/fp3.rs:41-48,
// :112-120), and the square roots built on it.  Same values as the reference's
// Karatsuba forms; here the nine-term convolution in Acc3 accumulators.
struct Fp3 {
    u64 a[3];
};
CH_DEV Fp3 fp3_one() {
    Fp3 r;
    r.a[0] = 1;
    r.a[1] = 0;
    r.a[2] = 0;
    return r;
}
CH_DEV Fp3 fp3_zero() {
    Fp3 r;
    r.a[0] = r.a[1] = r.a[2] = 0;
    return r;
}
// src/fp3.rs:169-211 (mul), :215-252 (square)
__device__ __noinline__ void fp3_mul_fn(const u64 x0, const u64 x1, const u64 x2, const u64 y0, const u64 y1, const u64 y2, u64* out) {
    const u64 y1s = fp_mul_small(y1, 7u), y2s = fp_mul_small(y2, 7u);
    Acc3 acc;
    acc.init2(x0, y0, x1, y2s);
    acc.mac(x2, y1s);
    out[0] = acc.reduce();
    acc.init2(x0, y1, x1, y0);
    acc.mac(x2, y2s);
    out[1] = acc.reduce();
    acc.init2(x0, y2, x1, y1);
    acc.mac(x2, y0);
    out[2] = acc.reduce();
}
CH_DEV Fp3 fp3_mul(const Fp3& x, const Fp3& y) {
    Fp3 r;
    fp3_mul_fn(x.a[0], x.a[1], x.a[2], y.a[0], y.a[1], y.a[2], r.a);
    return r;
}
CH_DEV Fp3 fp3_square(const Fp3& x) { return fp3_mul(x, x); }
CH_DEV bool fp3_eq(const Fp3& x, const Fp3& y) {
    return fp_canonical(x.a[0]) == fp_canonical(y.a[0]) && fp_canonical(x.a[1]) == fp_canonical(y.a[1]) &&
           fp_canonical(x.a[2]) == fp_canonical(y.a[2]);
}
CH_DEV Fp3 fp3_select(const Fp3& a, const Fp3& b, bool choice) {  // choice ? b : a
    Fp3 r;
#pragma unroll
    for (int i = 0; i < 3; i++) r.a[i] = choice ? b.a[i] : a.a[i];
    return r;
}
// Fp3::sqrt: Tonelli-Shanks for 2-adicity 32 in its constant-time form (src/fp3.rs:255-282).
// Returns s and *is_some = (s^2 == x).
CH_DEV Fp3 fp3_sqrt(const Fp3& x, bool* is_some) {
    // y = x^((t - 1) / 2), exponent 0x7ffffffe80000002fffffffc80000002fffffffe (src/fp3.rs:260-262)
    const u64 E[3] = {0x80000002fffffffeull, 0x80000002fffffffcull, 0x000000007ffffffeull};
    Fp3 y = fp3_one();
    for (int w = 2; w >= 0; w--) {
        for (int i = 63; i >= 0; i--) {
            y = fp3_square(y);
            if ((E[w] >> i) & 1) y = fp3_mul(y, x);  // fixed exponent: uniform branch
        }
    }
    Fp3 s = fp3_mul(x, y), t = fp3_mul(s, y);
    Fp3 z;
    z.a[0] = 2800184025912956819ull;  // TWO_ADIC_ROOT_OF_UNITY_P3, src/fp3.rs:33-37
    z.a[1] = 0;
    z.a[2] = 0;
    for (int k = 32; k >= 2; k--) {
        Fp3 b = t;
        for (int i = 0; i < k - 2; i++) b = fp3_square(b);
        const bool is_one = fp3_eq(b, fp3_one());
        const Fp3 new_s = fp3_mul(s, z);
        s = fp3_select(new_s, s, is_one);
        z = fp3_square(z);
        const Fp3 new_t = fp3_mul(t, z);
        t = fp3_select(new_t, t, is_one);
    }
    *is_some = fp3_eq(fp3_square(s), x);
    return s;
}

// (A0 + A1 Y) * f and (A0 + A1 Y) * (f Y) with Y^2 = X, A0 = (c0, c2, c4), A1 = (c1, c3, c5)
// (Fp6::mul_by_low_fp3 / mul_by_high_fp3, src/fp6.rs:469-647).
CH_DEV Fp6 fp6_mul_by_low_fp3(const Fp6& s, const Fp3& f) {
    Fp3 a0, a1;
    a0.a[0] = s.c[0]; a0.a[1] = s.c[2]; a0.a[2] = s.c[4];
    a1.a[0] = s.c[1]; a1.a[1] = s.c[3]; a1.a[2] = s.c[5];
    const Fp3 lo = fp3_mul(a0, f), hi = fp3_mul(a1, f);
    Fp6 r;
    r.c[0] = lo.a[0]; r.c[2] = lo.a[1]; r.c[4] = lo.a[2];
    r.c[1] = hi.a[0]; r.c[3] = hi.a[1]; r.c[5] = hi.a[2];
    return r;
}
CH_DEV Fp6 fp6_mul_by_high_fp3(const Fp6& s, const Fp3& f) {
    Fp3 a0, a1;
    a0.a[0] = s.c[0]; a0.a[1] = s.c[2]; a0.a[2] = s.c[4];
    a1.a[0] = s.c[1]; a1.a[1] = s.c[3]; a1.a[2] = s.c[5];
    const Fp3 d1 = fp3_mul(a1, f), d2 = fp3_mul(a0, f);
    Fp6 r;  // low half = X * (A1 f) = (7 d1[2], d1[0], d1[1]); high half = A0 f
    r.c[0] = fp_mul_small(d1.a[2], 7u); r.c[2] = d1.a[0]; r.c[4] = d1.a[1];
    r.c[1] = d2.a[0]; r.c[3] = d2.a[1]; r.c[5] = d2.a[2];
    return r;
}
CH_DEV bool fp6_eq(const Fp6& a, const Fp6& b) {
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) acc |= fp_canonical(a.c[i]) ^ fp_canonical(b.c[i]);
    return acc == 0;
}
// Fp6::sqrt, Algorithm 10 of eprint 2012/685 as in src/fp6.rs:922-964.
CH_DEV Fp6 fp6_sqrt(const Fp6& self, bool* is_some) {
    Fp3 E, F;
    E.a[0] = 17474221990273267221ull; E.a[1] = 11073203528037158631ull; E.a[2] = 1587630551273719300ull;
    F.a[0] = 16; F.a[1] = 18446744069414584320ull; F.a[2] = 0;
    Fp6 b = fp6_mul(self, fp6_frobenius(self));
    b = fp6_mul(b, fp6_frobenius_double(self));
    {   // b^0x3fffffffc0000000 (src/fp6.rs:947): 32 one bits, then 30 squarings
        const Fp6 base = b;
        Fp6 r = base;
        for (int i = 0; i < 31; i++) r = fp6_mul(fp6_square(r), base);
        for (int i = 0; i < 30; i++) r = fp6_square(r);
        b = r;
    }
    const Fp6 b2 = fp6_square(b);
    const Fp6 b_pow_p3 = fp6_frobenius_triple(b);
    const Fp6 check = fp6_mul(b_pow_p3, b);
    const Fp6 s = fp6_mul(b2, self);
    const Fp6 sf = fp6_mul_by_low_fp3(s, F);
    Fp3 s3, sf3;
    s3.a[0] = s.c[0]; s3.a[1] = s.c[2]; s3.a[2] = s.c[4];
    sf3.a[0] = sf.c[0]; sf3.a[1] = sf.c[2]; sf3.a[2] = sf.c[4];
    const bool is_one = fp6_eq(check, fp6_one());
    // the reference takes both Fp3 roots and selects; only the selected one is computed here (same value)
    bool some3;
    Fp3 x0 = fp3_sqrt(is_one ? s3 : sf3, &some3);
    if (!some3) x0 = fp3_zero();  // unwrap_or(Fp3::zero())
    Fp6 x = fp6_mul_by_low_fp3(b_pow_p3, x0);
    const Fp6 xe = fp6_mul_by_high_fp3(x, E);
    x = fp6_select(xe, x, is_one);
    *is_some = fp6_eq(fp6_square(x), self);
    return x;
}

// a^power for a 384-bit little-endian exponent: Fp6::exp / exp_vartime, src/fp6.rs:1053-1086 (the same value; the
// exponent is one value for the whole launch, so the branch on its bits is uniform).
CH_DEV Fp6 fp6_exp(const Fp6& a, const u64 (&power)[6]) {
    Fp6 res = fp6_one();
    for (int w = 5; w >= 0; w--) {
        for (int i = 63; i >= 0; i--) {
            res = fp6_square(res);
            if ((power[w] >> i) & 1) res = fp6_mul(res, a);
        }
    }
    return res;
}

// a * (3u + 1185) = a * 3B.                          (src/curve/mod.rs:54-106)
CH_DEV Fp6 fp6_mul_by_3b(const Fp6& a) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        Acc3 acc;
        if (i == 0)
            acc.init_small2(a.c[0], 1185u, a.c[5], 21u);
        else
            acc.init_small2(a.c[i], 1185u, a.c[i - 1], 3u);
        r.c[i] = acc.reduce();
    }
    return r;
}

}  // namespace cheetah
