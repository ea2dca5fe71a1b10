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

// Fp6 mul / square are ~450 / ~350 instructions each.  Inlined into every point
// formula the ladder loop would exceed the instruction cache, so by default they
// are real functions.  Their arguments and result cross the call as 6 x u64
// vectors: hipcc passes those in VGPRs, whereas a second struct argument goes
// through the stack (scratch memory).
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
CH_FP6_COEFFWISE2(fp6_add, fp_add)          // src/fp6.rs:994-1003
CH_FP6_COEFFWISE2(fp6_sub, fp_sub)          // src/fp6.rs:1007-1016
CH_FP6_COEFFWISE1(fp6_neg, fp_neg)          // src/fp6.rs:1020-1029
CH_FP6_COEFFWISE1(fp6_double, fp_double)    // src/fp6.rs:968-977
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
CH_DEV bool fp6_is_zero(const Fp6& a) {  // src/fp6.rs:362-369
    bool z = true;
#pragma unroll
    for (int i = 0; i < 6; i++) z = z && fp_is_zero(a.c[i]);
    return z;
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

CH_FP6_FN fp6v fp6_mul_fn(fp6v av, fp6v bv) {
    const Fp6 a = fp6_from_v(av), b = fp6_from_v(bv);
    u64 b7[6];
    b7[0] = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) b7[j] = fp_mul_small(b.c[j], 7u);
    Fp6 r;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        Acc3 acc;
        acc.init(a.c[0], b.c[k]);
#pragma unroll
        for (int i = 1; i < 6; i++) {
            if (i <= k)
                acc.mac(a.c[i], b.c[k - i]);
            else
                acc.mac(a.c[i], b7[k + 6 - i]);
        }
        r.c[k] = acc.reduce();
    }
    return fp6_to_v(r);
}
CH_DEV Fp6 fp6_mul(const Fp6& a, const Fp6& b) { return fp6_from_v(fp6_mul_fn(fp6_to_v(a), fp6_to_v(b))); }

// a^2 in Fp6.                                        (src/fp6.rs:838-920)
// 21 products: cross terms use the doubled operand d_i = 2 a_i, wrapped terms
// the scaled operand s_j = 7 a_j.
CH_FP6_FN fp6v fp6_square_fn(fp6v av) {
    const Fp6 a = fp6_from_v(av);
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
        bool started = false;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int j = i; j < 6; j++) {
                const bool low = (i + j == k), high = (i + j == k + 6);
                if (low || high) {
                    const u64 x = (i == j) ? a.c[i] : d[i];
                    const u64 y = high ? s[j] : a.c[j];
                    if (!started) {
                        acc.init(x, y);
                        started = true;
                    } else {
                        acc.mac(x, y);
                    }
                }
            }
        }
        r.c[k] = acc.reduce();
    }
    return fp6_to_v(r);
}
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
    acc.init(a.c[0], b.c[0]);
#pragma unroll
    for (int i = 1; i < 6; i++) acc.mac(a.c[i], fp_mul_small(b.c[6 - i], 7u));
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
