/*
 * cheetah_oracle.c -- CPU restatement of the toposware/cheetah hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cheetah_oracle.h).  Plain C11 + unsigned
 * __int128.  Each function restates the reference function cited beside it,
 * operation for operation (same Karatsuba blocks, same 8p offsets, same formula
 * sequences, same ladders, quirks Q1-Q4 of SURVEY.md), so raw projective
 * outputs coincide with the reference crate, not only the affine ones.
 *
 * Pinned by tests/test_oracle_kat.py against the reference's own known-answer
 * vectors (tests/golden/reference_kats.json) and the independent big-integer
 * model oracle/model.py.
 */
#include "cheetah_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define P_MOD 0xffffffff00000001ULL /* src/fp.rs:35 */
#define EPS 0xffffffffULL           /* src/fp.rs:42 */

/* ------------------------------------------------------------------ Fp ---- */

/* src/fp.rs:623-646 */
static inline u64 reduce_u128(u128 x) {
    u64 ab = (u64)x;
    u64 cd = (u64)(x >> 64);
    u64 c = (u64)(uint32_t)cd;
    u64 d = cd >> 32;
    u64 r0 = ab - d;
    u64 of = ab < d;
    r0 = r0 - EPS * of;
    u64 r1 = (c << 32) - c;
    u64 res = r0 + r1;
    of = res < r0;
    return res + EPS * of;
}

/* src/fp.rs:653-667 */
static inline u64 reduce_u96(u128 x) {
    u64 c = (u64)(uint32_t)(x >> 64);
    u64 r0 = (u64)x;
    u64 r1 = (c << 32) - c;
    u64 res = r0 + r1;
    u64 of = res < r0;
    return res + EPS * of;
}

/* src/fp.rs:136-141 */
static inline u64 fp_add(u64 a, u64 b) {
    u64 d0 = a + b;
    u64 of = d0 < a;
    u64 d1 = d0 + EPS * of;
    of = d1 < d0;
    return d1 + EPS * of;
}

/* src/fp.rs:145-150 (shl64_by_u32_with_carry: src/utils.rs) */
static inline u64 fp_double(u64 a) {
    u64 d0 = a << 1;
    u64 of = a >> 63;
    u64 d1 = d0 + EPS * of;
    of = d1 < d0;
    return d1 + EPS * of;
}

/* src/fp.rs:154-159 */
static inline u64 fp_triple(u64 a) {
    u128 t = a;
    t = t + (t << 1);
    return reduce_u96(t);
}

/* src/fp.rs:163-168 */
static inline u64 fp_sub(u64 a, u64 b) {
    u64 d0 = a - b;
    u64 of = a < b;
    u64 d1 = d0 - EPS * of;
    of = d0 < EPS * of;
    return d1 - EPS * of;
}

/* src/fp.rs:172-174 */
static inline u64 fp_neg(u64 a) { return fp_sub(0, a); }

/* src/fp.rs:178-182 */
static inline u64 fp_mul(u64 a, u64 b) { return reduce_u128((u128)a * (u128)b); }

/* src/fp.rs:186-190 */
static inline u64 fp_mul_by_u32(u64 a, uint32_t b) { return reduce_u96((u128)a * (u128)b); }

/* src/fp.rs:194-196 */
static inline u64 fp_square(u64 a) { return fp_mul(a, a); }

static inline u64 fp_canonical(u64 a) { return a % P_MOD; } /* src/fp.rs:125-127 */

static inline u64 fp_sqn(u64 a, int n) { /* src/utils.rs:40-44 square_assign_multi */
    for (int i = 0; i < n; i++) a = fp_square(a);
    return a;
}

/* src/fp.rs:335-359 */
static int fp_invert(u64 self, u64 *out) {
    u64 t2 = fp_square(self);
    u64 t3 = fp_mul(t2, self);
    u64 t0 = fp_square(t3);
    t0 = fp_square(t0);
    t2 = fp_mul(t2, t0);
    t3 = fp_mul(t3, t0);
    t0 = fp_square(t3);
    t0 = fp_sqn(t0, 3);
    t2 = fp_mul(t2, t0);
    t3 = fp_mul(t3, t0);
    t0 = fp_square(t3);
    t0 = fp_sqn(t0, 7);
    t2 = fp_mul(t2, t0);
    t0 = fp_mul(t0, t3);
    t0 = fp_sqn(t0, 16);
    t2 = fp_mul(t2, t0);
    t0 = fp_square(t2);
    t0 = fp_sqn(t0, 31);
    t0 = fp_mul(t0, t2);
    t0 = fp_mul(t0, self);
    *out = t0;
    return fp_canonical(self) != 0;
}

u64 ora_fp_add(u64 a, u64 b) { return fp_add(a, b); }
u64 ora_fp_sub(u64 a, u64 b) { return fp_sub(a, b); }
u64 ora_fp_neg(u64 a) { return fp_neg(a); }
u64 ora_fp_double(u64 a) { return fp_double(a); }
u64 ora_fp_triple(u64 a) { return fp_triple(a); }
u64 ora_fp_mul(u64 a, u64 b) { return fp_mul(a, b); }
u64 ora_fp_square(u64 a) { return fp_square(a); }
u64 ora_fp_mul_by_u32(u64 a, uint32_t b) { return fp_mul_by_u32(a, b); }
u64 ora_fp_canonical(u64 a) { return fp_canonical(a); }
int ora_fp_invert(u64 a, u64 *out) {
    int s = fp_invert(a, out);
    if (!s) *out = 0;
    return s;
}
/* Fp::exp, src/fp.rs:304-315 (exp_vartime, :323-333, returns the same element) */
static u64 fp_exp(u64 self, u64 power) {
    u64 res = 1;
    for (int i = 63; i >= 0; i--) {
        res = fp_square(res);
        u64 tmp = fp_mul(res, self);
        if ((power >> i) & 1) res = tmp; /* conditional_assign */
    }
    return res;
}
u64 ora_fp_exp(u64 a, u64 power) { return fp_canonical(fp_exp(a, power)); }
/* Fp::sqrt, src/fp.rs:199-229: Tonelli-Shanks, constant-time form; returns is_some */
int ora_fp_sqrt(u64 self, u64 *out) {
    u64 y = fp_exp(self, 0x7fffffffULL);
    u64 s = fp_mul(self, y);
    u64 t = fp_mul(s, y);
    u64 z = 1753635133440165772ULL; /* TWO_ADIC_ROOT_OF_UNITY, src/fp.rs:48 */
    for (int k = 32; k >= 2; k--) {
        u64 b = fp_sqn(t, k - 2);
        int is_one = fp_canonical(b) == 1;
        u64 new_s = fp_mul(s, z);
        s = is_one ? s : new_s;
        z = fp_square(z);
        u64 new_t = fp_mul(t, z);
        t = is_one ? t : new_t;
    }
    int some = fp_canonical(fp_square(s)) == fp_canonical(self);
    *out = some ? fp_canonical(s) : 0;
    return some;
}
u64 ora_reduce_u128(u64 lo, u64 hi) { return reduce_u128(((u128)hi << 64) | lo); }
u64 ora_reduce_u96(u64 lo, uint32_t hi) { return reduce_u96(((u128)hi << 64) | lo); }

/* ----------------------------------------------------------------- Fp6 ---- */

typedef struct {
    u64 c[6];
} fp6;

static const fp6 FP6_ZERO = {{0, 0, 0, 0, 0, 0}};
static const fp6 FP6_ONE = {{1, 0, 0, 0, 0, 0}};

#define BETA ((u128)7)                                            /* src/fp6.rs:40 */
#define OFF8P ((((u128)0x7ULL) << 64) | (u128)0xfffffff800000008ULL) /* 0x7fffffff800000008 = 8p, src/fp6.rs:760 */

static inline fp6 fp6_load(const u64 *a) {
    fp6 r;
    memcpy(r.c, a, sizeof r.c);
    return r;
}
static inline void fp6_store(u64 *o, fp6 a) { memcpy(o, a.c, sizeof a.c); }

#define FP6_COEFFWISE(name, op)                                       \
    static inline fp6 name(fp6 a, fp6 b) {                            \
        fp6 r;                                                        \
        for (int i = 0; i < 6; i++) r.c[i] = op(a.c[i], b.c[i]);      \
        return r;                                                     \
    }
FP6_COEFFWISE(fp6_add, fp_add) /* src/fp6.rs:994-1003 */
FP6_COEFFWISE(fp6_sub, fp_sub) /* src/fp6.rs:1007-1016 */

static inline fp6 fp6_neg(fp6 a) { /* src/fp6.rs:1020-1029 */
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_neg(a.c[i]);
    return r;
}
static inline fp6 fp6_double(fp6 a) { /* src/fp6.rs:968-977 */
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_double(a.c[i]);
    return r;
}
static inline fp6 fp6_triple(fp6 a) { /* src/fp6.rs:981-990 */
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_triple(a.c[i]);
    return r;
}
static inline fp6 fp6_mul_by_u32(fp6 a, uint32_t b) { /* src/fp6.rs:428-444 */
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul_by_u32(a.c[i], b);
    return r;
}
static inline fp6 fp6_mul_by_fp(fp6 a, u64 b) { /* src/fp6.rs:448-464 */
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_mul(a.c[i], b);
    return r;
}
static inline int fp6_is_zero(fp6 a) { /* src/fp6.rs:362-369 */
    int z = 1;
    for (int i = 0; i < 6; i++) z &= (fp_canonical(a.c[i]) == 0);
    return z;
}
static inline fp6 fp6_canonical(fp6 a) { /* src/fp6.rs:373-382 */
    fp6 r;
    for (int i = 0; i < 6; i++) r.c[i] = fp_canonical(a.c[i]);
    return r;
}
static inline int fp6_eq(fp6 a, fp6 b) { /* ct_eq on canonical values */
    int e = 1;
    for (int i = 0; i < 6; i++) e &= (fp_canonical(a.c[i]) == fp_canonical(b.c[i]));
    return e;
}
/* subtle conditional_select(a, b, choice): choice ? b : a   src/fp6.rs:247-258 */
static inline fp6 fp6_select(fp6 a, fp6 b, int choice) { return choice ? b : a; }

/* One cubic Karatsuba block of src/fp6.rs:687-816: (x0,x1,x2)*(y0,y1,y2) in
 * Fp[X]/(X^3-7), returning the three unreduced u128 coefficients d0,d1,d2. */
static inline void cubic_block(u64 x0, u64 x1, u64 x2, u64 y0, u64 y1, u64 y2, u128 *d0, u128 *d1, u128 *d2) {
    u128 t0 = fp_mul(x0, y0);
    u128 t1 = fp_mul(x1, y1);
    u128 t2 = fp_mul(x2, y2);
    u128 s12 = fp_mul(fp_add(x1, x2), fp_add(y1, y2));
    u128 s01 = fp_mul(fp_add(x0, x1), fp_add(y0, y1));
    u128 s02 = fp_mul(fp_add(x0, x2), fp_add(y0, y2));

    u128 e0 = t1 + t2;
    e0 = s12 + OFF8P - e0;
    e0 = e0 * BETA;
    e0 = e0 + t0;

    u128 e1 = t2 * BETA;
    u128 tmp = t0 + t1;
    e1 = e1 + OFF8P - tmp;
    e1 = e1 + s01;

    u128 e2 = t0 + t2;
    e2 = t1 + OFF8P - e2;
    e2 = e2 + s02;

    *d0 = e0;
    *d1 = e1;
    *d2 = e2;
}

/* src/fp6.rs:651-834 */
static fp6 fp6_mul(fp6 a, fp6 b) {
    u128 d00, d01, d02, d10, d11, d12, d20, d21, d22, d30, d31, d32;
    /* a_0 * b_0 */
    cubic_block(a.c[0], a.c[2], a.c[4], b.c[0], b.c[2], b.c[4], &d00, &d01, &d02);
    /* a_1 * b_1 */
    cubic_block(a.c[1], a.c[3], a.c[5], b.c[1], b.c[3], b.c[5], &d10, &d11, &d12);
    /* a_0 * b_1 */
    cubic_block(a.c[0], a.c[2], a.c[4], b.c[1], b.c[3], b.c[5], &d20, &d21, &d22);
    /* a_1 * b_0 */
    cubic_block(a.c[1], a.c[3], a.c[5], b.c[0], b.c[2], b.c[4], &d30, &d31, &d32);

    fp6 r;
    r.c[0] = reduce_u96(d00 + d12 * BETA);
    r.c[2] = reduce_u96(d01 + d10);
    r.c[4] = reduce_u96(d02 + d11);
    r.c[1] = reduce_u96(d20 + d30);
    r.c[3] = reduce_u96(d21 + d31);
    r.c[5] = reduce_u96(d22 + d32);
    return r;
}

/* src/fp6.rs:838-920 */
static fp6 fp6_square(fp6 s) {
    u128 aa = fp_square(s.c[0]);
    u128 ab = fp_mul(s.c[0], s.c[1]);
    u128 ac = fp_mul(s.c[0], s.c[2]);
    u128 ad = fp_mul(s.c[0], s.c[3]);
    u128 ae = fp_mul(s.c[0], s.c[4]);
    u128 af = fp_mul(s.c[0], s.c[5]);
    u128 bb = fp_square(s.c[1]);
    u128 bc = fp_mul(s.c[1], s.c[2]);
    u128 bd = fp_mul(s.c[1], s.c[3]);
    u128 be = fp_mul(s.c[1], s.c[4]);
    u128 bf = fp_mul(s.c[1], s.c[5]);
    u128 cc = fp_square(s.c[2]);
    u128 cd = fp_mul(s.c[2], s.c[3]);
    u128 ce = fp_mul(s.c[2], s.c[4]);
    u128 cf = fp_mul(s.c[2], s.c[5]);
    u128 dd = fp_square(s.c[3]);
    u128 de = fp_mul(s.c[3], s.c[4]);
    u128 df = fp_mul(s.c[3], s.c[5]);
    u128 ee = fp_square(s.c[4]);
    u128 ef = fp_mul(s.c[4], s.c[5]);
    u128 ff = fp_square(s.c[5]);

    fp6 r;
    u128 c0 = bf + ce;
    c0 = c0 << 1;
    c0 = c0 + dd;
    c0 = c0 * BETA;
    c0 = c0 + aa;
    r.c[0] = reduce_u96(c0);

    u128 c1 = cf + de;
    c1 = c1 * BETA;
    c1 = c1 + ab;
    c1 = c1 << 1;
    r.c[1] = reduce_u96(c1);

    u128 c2 = df << 1;
    c2 = c2 + ee;
    c2 = c2 * BETA;
    u128 t2 = ac << 1;
    c2 = c2 + t2;
    c2 = c2 + bb;
    r.c[2] = reduce_u96(c2);

    u128 c3 = ef * BETA;
    c3 = c3 + ad;
    c3 = c3 + bc;
    c3 = c3 << 1;
    r.c[3] = reduce_u96(c3);

    u128 t4 = ff * BETA;
    u128 c4 = ae + bd;
    c4 = c4 << 1;
    c4 = c4 + cc;
    c4 = c4 + t4;
    r.c[4] = reduce_u96(c4);

    u128 c5 = af + be;
    c5 = c5 + cd;
    c5 = c5 << 1;
    r.c[5] = reduce_u96(c5);
    return r;
}

/* src/fp6.rs:1088-1097 */
static fp6 fp6_frobenius(fp6 a) {
    fp6 r;
    r.c[0] = a.c[0];
    r.c[1] = fp_mul(18446744065119617026ULL, a.c[1]);
    r.c[2] = fp_mul(18446744065119617025ULL, a.c[2]);
    r.c[3] = fp_neg(a.c[3]);
    r.c[4] = fp_mul_by_u32(a.c[4], 4294967295U);
    r.c[5] = fp_mul(4294967296ULL, a.c[5]);
    return r;
}
/* src/fp6.rs:1101-1110 */
static fp6 fp6_frobenius_double(fp6 a) {
    fp6 r;
    r.c[0] = a.c[0];
    r.c[1] = fp_mul(18446744065119617025ULL, a.c[1]);
    r.c[2] = fp_mul_by_u32(a.c[2], 4294967295U);
    r.c[3] = a.c[3];
    r.c[4] = fp_mul(18446744065119617025ULL, a.c[4]);
    r.c[5] = fp_mul_by_u32(a.c[5], 4294967295U);
    return r;
}
/* src/fp6.rs:1114-1123 */
static fp6 fp6_frobenius_triple(fp6 a) {
    fp6 r;
    r.c[0] = a.c[0];
    r.c[1] = fp_neg(a.c[1]);
    r.c[2] = a.c[2];
    r.c[3] = fp_neg(a.c[3]);
    r.c[4] = a.c[4];
    r.c[5] = fp_neg(a.c[5]);
    return r;
}

/* src/fp6.rs:1035-1049; out untouched if none */
static int fp6_invert(fp6 self, fp6 *out) {
    fp6 t0 = fp6_frobenius(self);
    fp6 t1 = fp6_frobenius_triple(self);
    fp6 t2 = fp6_mul(t0, t1);
    fp6 t3 = fp6_frobenius(t2);
    fp6 t4 = fp6_frobenius_double(t2);
    fp6 t5 = fp6_mul(t4, t3);
    t5 = fp6_mul(t5, t0);
    u64 inv = fp6_mul(t5, self).c[0];
    u64 t;
    if (!fp_invert(inv, &t)) return 0;
    *out = fp6_mul_by_fp(t5, t);
    return 1;
}

/* src/curve/mod.rs:54-106 (B3.c0 = 1185) */
static fp6 mul_by_3b(fp6 a) {
    const u64 B3C0 = 1185;
    u128 aa = fp_mul(a.c[0], B3C0);
    u128 ab = a.c[0];
    ab = ab + (ab << 1);
    u128 ba = fp_mul(a.c[1], B3C0);
    u128 bb = a.c[1];
    bb = bb + (bb << 1);
    u128 ca = fp_mul(a.c[2], B3C0);
    u128 cb = a.c[2];
    cb = cb + (cb << 1);
    u128 da = fp_mul(a.c[3], B3C0);
    u128 db = a.c[3];
    db = db + (db << 1);
    u128 ea = fp_mul(a.c[4], B3C0);
    u128 eb = a.c[4];
    eb = eb + (eb << 1);
    u128 fa = fp_mul(a.c[5], B3C0);
    u128 fb = a.c[5];
    fb = fb + (fb << 1);
    fp6 r;
    r.c[0] = reduce_u96(fb * (u128)7 + aa);
    r.c[1] = reduce_u96(ab + ba);
    r.c[2] = reduce_u96(bb + ca);
    r.c[3] = reduce_u96(cb + da);
    r.c[4] = reduce_u96(db + ea);
    r.c[5] = reduce_u96(eb + fa);
    return r;
}

void ora_fp6_add(const u64 *a, const u64 *b, u64 *o) { fp6_store(o, fp6_add(fp6_load(a), fp6_load(b))); }
void ora_fp6_sub(const u64 *a, const u64 *b, u64 *o) { fp6_store(o, fp6_sub(fp6_load(a), fp6_load(b))); }
void ora_fp6_neg(const u64 *a, u64 *o) { fp6_store(o, fp6_neg(fp6_load(a))); }
void ora_fp6_double(const u64 *a, u64 *o) { fp6_store(o, fp6_double(fp6_load(a))); }
void ora_fp6_triple(const u64 *a, u64 *o) { fp6_store(o, fp6_triple(fp6_load(a))); }
void ora_fp6_mul(const u64 *a, const u64 *b, u64 *o) { fp6_store(o, fp6_mul(fp6_load(a), fp6_load(b))); }
void ora_fp6_square(const u64 *a, u64 *o) { fp6_store(o, fp6_square(fp6_load(a))); }
void ora_fp6_mul_by_u32(const u64 *a, uint32_t b, u64 *o) { fp6_store(o, fp6_mul_by_u32(fp6_load(a), b)); }
void ora_fp6_mul_by_fp(const u64 *a, u64 b, u64 *o) { fp6_store(o, fp6_mul_by_fp(fp6_load(a), b)); }
void ora_fp6_frobenius(const u64 *a, u64 *o) { fp6_store(o, fp6_frobenius(fp6_load(a))); }
void ora_fp6_frobenius_double(const u64 *a, u64 *o) { fp6_store(o, fp6_frobenius_double(fp6_load(a))); }
void ora_fp6_frobenius_triple(const u64 *a, u64 *o) { fp6_store(o, fp6_frobenius_triple(fp6_load(a))); }
int ora_fp6_invert(const u64 *a, u64 *o) {
    fp6 r = FP6_ZERO;
    int s = fp6_invert(fp6_load(a), &r);
    fp6_store(o, s ? r : FP6_ZERO);
    return s;
}
void ora_fp6_canonical(const u64 *a, u64 *o) { fp6_store(o, fp6_canonical(fp6_load(a))); }
int ora_fp6_is_zero(const u64 *a) { return fp6_is_zero(fp6_load(a)); }
void ora_mul_by_3b(const u64 *a, u64 *o) { fp6_store(o, mul_by_3b(fp6_load(a))); }

/* ------------------------------------------------------------ recoders ---- */

/* src/scalar.rs:531-548 */
void ora_bytes_to_radix_16(const uint8_t *bytes, int8_t *result) {
    for (int i = 0; i < 32; i++) {
        result[2 * i] = (int8_t)(bytes[i] & 0xf);
        result[2 * i + 1] = (int8_t)((bytes[i] >> 4) & 0xf);
    }
    for (int i = 0; i < 63; i++) {
        int8_t carry = (int8_t)((result[i] + 8) >> 4);
        result[i] = (int8_t)(result[i] - (carry << 4));
        result[i + 1] = (int8_t)(result[i + 1] + carry);
    }
}

/* src/scalar.rs:555-576 */
void ora_bytes_to_radix_256(const uint8_t *bytes, int8_t *out) {
    int16_t r[32];
    for (int i = 0; i < 32; i++) r[i] = (int16_t)bytes[i];
    for (int i = 0; i < 31; i++) {
        int16_t carry = (int16_t)((r[i] + 128) >> 8);
        r[i] = (int16_t)(r[i] - (carry << 8));
        r[i + 1] = (int16_t)(r[i + 1] + carry);
    }
    for (int i = 0; i < 32; i++) out[i] = (int8_t)r[i];
}

/* src/scalar.rs:585-643 */
void ora_bytes_to_wnaf_vartime(const uint8_t *bytes, int w, int8_t *naf) {
    memset(naf, 0, 256);
    u64 x[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u64 v = 0;
        for (int j = 7; j >= 0; j--) v = (v << 8) | bytes[8 * i + j];
        x[i] = v;
    }
    u64 width = 1ULL << w;
    u64 window_mask = width - 1;
    size_t pos = 0;
    u64 carry = 0;
    while (pos < 256) {
        size_t idx = pos / 64, bit = pos % 64;
        u64 bit_buf;
        if (bit < 64 - (size_t)w)
            bit_buf = x[idx] >> bit;
        else
            bit_buf = (x[idx] >> bit) | (x[1 + idx] << (64 - bit));
        u64 window = carry + (bit_buf & window_mask);
        if ((window & 1) == 0) {
            pos += 1;
            continue;
        }
        if (window < width / 2) {
            carry = 0;
            naf[pos] = (int8_t)window;
        } else {
            carry = 1;
            naf[pos] = (int8_t)((int8_t)window - (int8_t)width); /* wrapping_sub on i8 */
        }
        pos += (size_t)w;
    }
}

/* -------------------------------------------------------------- points ---- */

typedef struct {
    fp6 x, y;
    uint8_t inf;
} affine;
typedef struct {
    fp6 x, y, z;
} proj; /* also used for jacobian */
typedef struct {
    fp6 x, y, z, z4;
} mjac;

static inline affine affine_identity(void) { /* src/curve/affine.rs:224-230 */
    affine a = {FP6_ZERO, FP6_ONE, 1};
    return a;
}
static inline proj proj_identity(void) { /* src/curve/projective.rs:204-210; jacobian identical */
    proj p = {FP6_ZERO, FP6_ONE, FP6_ZERO};
    return p;
}
static inline affine affine_neg(affine a) { /* src/curve/affine.rs:164-175 */
    affine r;
    r.x = a.x;
    r.y = a.inf ? FP6_ONE : fp6_neg(a.y);
    r.inf = a.inf;
    return r;
}
static inline proj proj_from_affine(affine a) { /* src/curve/projective.rs:102-110 */
    proj p = {a.x, a.y, a.inf ? FP6_ZERO : FP6_ONE};
    return p;
}
static inline mjac mjac_from_affine(affine a) { /* src/curve/jacobian.rs:970-979 */
    mjac p = {a.x, a.y, a.inf ? FP6_ZERO : FP6_ONE, a.inf ? FP6_ZERO : FP6_ONE};
    return p;
}
static inline proj proj_load(const u64 *p) {
    proj r = {fp6_load(p), fp6_load(p + 6), fp6_load(p + 12)};
    return r;
}
static inline void proj_store(u64 *o, proj p) {
    fp6_store(o, p.x);
    fp6_store(o + 6, p.y);
    fp6_store(o + 12, p.z);
}
static inline affine affine_load(const u64 *xy, uint8_t inf) {
    affine a = {fp6_load(xy), fp6_load(xy + 6), (uint8_t)(inf != 0)};
    return a;
}
static inline void affine_store(u64 *xy, uint8_t *inf, affine a) {
    fp6_store(xy, fp6_canonical(a.x));
    fp6_store(xy + 6, fp6_canonical(a.y));
    if (inf) *inf = a.inf;
}

/* src/curve/projective.rs:349-393 */
static proj proj_double(proj s) {
    fp6 x2 = fp6_square(s.x);
    fp6 z2 = fp6_square(s.z);
    fp6 a = fp6_double(x2);
    a = fp6_add(a, x2);
    a = fp6_add(a, z2);
    fp6 b = fp6_mul(s.y, s.z);
    fp6 t3 = fp6_mul(s.y, b);
    fp6 c = fp6_mul(t3, s.x);
    fp6 c4 = fp6_mul_by_u32(c, 4);
    fp6 d = fp6_double(c4);
    fp6 t = fp6_square(a);
    d = fp6_sub(t, d);
    fp6 x3 = fp6_mul(b, d);
    x3 = fp6_double(x3);
    fp6 y3 = fp6_sub(c4, d);
    y3 = fp6_mul(a, y3);
    t3 = fp6_square(t3);
    t3 = fp6_mul_by_u32(t3, 8);
    y3 = fp6_sub(y3, t3);
    fp6 z3 = fp6_square(b);
    z3 = fp6_mul(z3, b);
    z3 = fp6_mul_by_u32(z3, 8);
    proj r = {x3, y3, z3};
    if (fp6_is_zero(s.z)) return proj_identity();
    return r;
}

/* src/curve/projective.rs:400-443 */
static proj proj_double_unchecked(proj s) {
    fp6 x2 = fp6_square(s.x);
    fp6 z2 = fp6_square(s.z);
    fp6 a = fp6_double(x2);
    a = fp6_add(a, x2);
    a = fp6_add(a, z2);
    fp6 b = fp6_mul(s.y, s.z);
    fp6 t3 = fp6_mul(s.y, b);
    fp6 c = fp6_mul(t3, s.x);
    fp6 d = fp6_double(c);
    fp6 c4 = fp6_double(d);
    d = fp6_double(c4);
    fp6 t = fp6_square(a);
    d = fp6_sub(t, d);
    fp6 x3 = fp6_mul(b, d);
    x3 = fp6_double(x3);
    fp6 y3 = fp6_sub(c4, d);
    y3 = fp6_mul(a, y3);
    t3 = fp6_double(t3);
    t3 = fp6_square(t3);
    t3 = fp6_double(t3);
    y3 = fp6_sub(y3, t3);
    fp6 z3 = fp6_double(b);
    z3 = fp6_square(z3);
    z3 = fp6_mul(z3, b);
    z3 = fp6_double(z3);
    proj r = {x3, y3, z3};
    return r;
}

/* src/curve/projective.rs:456-515 */
static proj proj_add(proj s, proj r) {
    fp6 t0 = fp6_mul(s.x, r.x);
    fp6 t1 = fp6_mul(s.y, r.y);
    fp6 t2 = fp6_mul(s.z, r.z);
    fp6 t3 = fp6_add(s.x, s.y);
    fp6 t4 = fp6_add(r.x, r.y);
    t3 = fp6_mul(t3, t4);
    t4 = fp6_add(t0, t1);
    t3 = fp6_sub(t3, t4);
    t4 = fp6_add(s.x, s.z);
    fp6 t5 = fp6_add(r.x, r.z);
    t4 = fp6_mul(t4, t5);
    t5 = fp6_add(t0, t2);
    t4 = fp6_sub(t4, t5);
    t5 = fp6_add(s.y, s.z);
    fp6 x3 = fp6_add(r.y, r.z);
    t5 = fp6_mul(t5, x3);
    x3 = fp6_add(t1, t2);
    t5 = fp6_sub(t5, x3);
    x3 = mul_by_3b(t2);
    fp6 z3 = fp6_add(x3, t4);
    x3 = fp6_sub(t1, z3);
    z3 = fp6_add(t1, z3);
    fp6 y3 = fp6_mul(x3, z3);
    t1 = fp6_double(t0);
    t1 = fp6_add(t1, t0);
    t4 = mul_by_3b(t4);
    t1 = fp6_add(t1, t2);
    t2 = fp6_sub(t0, t2);
    t4 = fp6_add(t4, t2);
    t0 = fp6_mul(t1, t4);
    y3 = fp6_add(y3, t0);
    t0 = fp6_mul(t5, t4);
    x3 = fp6_mul(t3, x3);
    x3 = fp6_sub(x3, t0);
    t0 = fp6_mul(t3, t1);
    z3 = fp6_mul(t5, z3);
    z3 = fp6_add(z3, t0);
    proj o = {x3, y3, z3};
    return o;
}

/* src/curve/projective.rs:518-569 */
static proj proj_add_mixed(proj s, affine r) {
    fp6 t0 = fp6_mul(s.x, r.x);
    fp6 t1 = fp6_mul(s.y, r.y);
    fp6 t3 = fp6_add(r.x, r.y);
    fp6 t4 = fp6_add(s.x, s.y);
    t3 = fp6_mul(t3, t4);
    t4 = fp6_add(t0, t1);
    t3 = fp6_sub(t3, t4);
    t4 = fp6_mul(r.x, s.z);
    t4 = fp6_add(t4, s.x);
    fp6 t5 = fp6_mul(r.y, s.z);
    t5 = fp6_add(t5, s.y);
    fp6 x3 = mul_by_3b(s.z);
    fp6 z3 = fp6_add(x3, t4);
    x3 = fp6_sub(t1, z3);
    z3 = fp6_add(t1, z3);
    fp6 y3 = fp6_mul(x3, z3);
    t1 = fp6_double(t0);
    t1 = fp6_add(t1, t0);
    t4 = mul_by_3b(t4);
    t1 = fp6_add(t1, s.z);
    fp6 t2 = fp6_sub(t0, s.z);
    t4 = fp6_add(t4, t2);
    t0 = fp6_mul(t1, t4);
    y3 = fp6_add(y3, t0);
    t0 = fp6_mul(t5, t4);
    x3 = fp6_mul(t3, x3);
    x3 = fp6_sub(x3, t0);
    t0 = fp6_mul(t3, t1);
    z3 = fp6_mul(t5, z3);
    z3 = fp6_add(z3, t0);
    proj o = {x3, y3, z3};
    return r.inf ? s : o;
}

/* src/curve/projective.rs:576-610 */
static proj proj_add_mixed_unchecked(proj s, affine r) {
    fp6 t0 = fp6_mul(r.y, s.z);
    t0 = fp6_sub(t0, s.y);
    fp6 t1 = fp6_mul(r.x, s.z);
    t1 = fp6_sub(t1, s.x);
    fp6 t2 = fp6_square(t0);
    t2 = fp6_mul(t2, s.z);
    fp6 t3 = fp6_square(t1);
    fp6 t4 = fp6_mul(t1, t3);
    t2 = fp6_sub(t2, t4);
    fp6 t5 = fp6_mul(t3, s.x);
    fp6 t6 = fp6_double(t5);
    t2 = fp6_sub(t2, t6);
    fp6 x3 = fp6_mul(t1, t2);
    fp6 y3 = fp6_sub(t5, t2);
    y3 = fp6_mul(y3, t0);
    fp6 t7 = fp6_mul(t4, s.y);
    y3 = fp6_sub(y3, t7);
    fp6 z3 = fp6_mul(t4, s.z);
    proj o = {x3, y3, z3};
    return r.inf ? s : o;
}

/* src/curve/affine.rs:64-78 */
static affine affine_from_proj(proj p) {
    fp6 zinv = FP6_ZERO;
    int some = fp6_invert(p.z, &zinv);
    if (!some) zinv = FP6_ZERO;
    affine a;
    a.x = fp6_mul(p.x, zinv);
    a.y = fp6_mul(p.y, zinv);
    a.inf = 0;
    if (fp6_is_zero(zinv)) return affine_identity();
    return a;
}

/* src/curve/affine.rs:86-101 */
static affine affine_from_jac(proj p) {
    fp6 z3 = fp6_mul(fp6_square(p.z), p.z);
    fp6 zinv3 = FP6_ZERO;
    if (!fp6_invert(z3, &zinv3)) zinv3 = FP6_ZERO;
    affine a;
    a.x = fp6_mul(fp6_mul(p.x, zinv3), p.z);
    a.y = fp6_mul(p.y, zinv3);
    a.inf = 0;
    if (fp6_is_zero(zinv3)) return affine_identity();
    return a;
}

/* src/curve/affine.rs:109-124 */
static affine affine_from_mjac(mjac p) {
    fp6 zinv4 = FP6_ZERO;
    if (!fp6_invert(p.z4, &zinv4)) zinv4 = FP6_ZERO;
    fp6 zinv3 = fp6_mul(zinv4, p.z);
    affine a;
    a.x = fp6_mul(fp6_mul(p.x, zinv3), p.z);
    a.y = fp6_mul(p.y, zinv3);
    a.inf = 0;
    if (fp6_is_zero(zinv4)) return affine_identity();
    return a;
}

static const fp6 CURVE_B = {{395, 1, 0, 0, 0, 0}}; /* src/curve/mod.rs:42-49 */

/* src/curve/projective.rs:940-946 */
static int proj_is_on_curve(proj p) {
    fp6 lhs = fp6_mul(fp6_square(p.y), p.z);
    fp6 z2 = fp6_square(p.z);
    fp6 rhs = fp6_add(fp6_add(fp6_mul(fp6_square(p.x), p.x), fp6_mul(p.x, z2)), fp6_mul(fp6_mul(z2, p.z), CURVE_B));
    return fp6_eq(lhs, rhs) | fp6_is_zero(p.z);
}

/* src/curve/jacobian.rs:930-938 */
static int jac_is_on_curve(proj p) {
    fp6 z2 = fp6_square(p.z);
    fp6 z4 = fp6_square(z2);
    fp6 rhs = fp6_add(fp6_add(fp6_mul(fp6_square(p.x), p.x), fp6_mul(p.x, z4)), fp6_mul(fp6_mul(z2, z4), CURVE_B));
    return fp6_eq(fp6_square(p.y), rhs) | fp6_is_zero(p.z);
}

/* src/curve/jacobian.rs:409-439 */
static proj jac_double_unchecked(proj s) {
    fp6 x_sq = fp6_square(s.x);
    fp6 y_sq = fp6_square(s.y);
    fp6 y_sq2 = fp6_square(y_sq);
    fp6 z_sq = fp6_square(s.z);
    fp6 z_sq2 = fp6_square(z_sq);
    fp6 a = fp6_mul_by_u32(y_sq, 4);
    a = fp6_mul(a, s.x);
    fp6 b = fp6_triple(x_sq);
    b = fp6_add(b, z_sq2);
    fp6 x3 = fp6_square(b);
    x3 = fp6_sub(x3, fp6_double(a));
    fp6 y3 = fp6_sub(a, x3);
    y3 = fp6_mul(y3, b);
    y3 = fp6_sub(y3, fp6_mul_by_u32(y_sq2, 8));
    fp6 z3 = fp6_double(s.y);
    z3 = fp6_mul(z3, s.z);
    proj o = {x3, y3, z3};
    return o;
}

static proj jac_add_mixed_unchecked(proj s, affine r);
static inline proj jac_identity(void) { /* JacobianPoint::identity() = (1, 1, 0), src/curve/jacobian.rs:223-229 */
    proj p = {FP6_ONE, FP6_ONE, FP6_ZERO};
    return p;
}

/* src/curve/jacobian.rs:368-402 */
static proj jac_double(proj s) {
    proj o = jac_double_unchecked(s); /* identical arithmetic sequence */
    if (fp6_is_zero(s.z)) return jac_identity();
    return o;
}

/* ConstantTimeEq for JacobianPoint, src/curve/jacobian.rs:134-154 */
static int jac_eq(proj s, proj o) {
    fp6 z1_squared = fp6_square(s.z);
    fp6 z2_squared = fp6_square(o.z);
    fp6 x1 = fp6_mul(s.x, z2_squared);
    fp6 x2 = fp6_mul(o.x, z1_squared);
    fp6 y1 = fp6_mul(fp6_mul(s.y, z2_squared), o.z);
    fp6 y2 = fp6_mul(fp6_mul(o.y, z1_squared), s.z);
    int sz = fp6_is_zero(s.z), oz = fp6_is_zero(o.z);
    return (sz & oz) | (!sz & !oz & fp6_eq(x1, x2) & fp6_eq(y1, y2));
}
/* ConstantTimeEq for ProjectivePoint, src/curve/projective.rs:118-135 */
static int proj_eq(proj s, proj o) {
    fp6 x1 = fp6_mul(s.x, o.z);
    fp6 x2 = fp6_mul(o.x, s.z);
    fp6 y1 = fp6_mul(s.y, o.z);
    fp6 y2 = fp6_mul(o.y, s.z);
    int sz = fp6_is_zero(s.z), oz = fp6_is_zero(o.z);
    return (sz & oz) | (!sz & !oz & fp6_eq(x1, x2) & fp6_eq(y1, y2));
}
/* ConstantTimeEq for AffinePoint, src/curve/affine.rs:132-144 */
static int affine_eq(affine a, affine b) {
    return (a.inf & b.inf) | (!a.inf & !b.inf & fp6_eq(a.x, b.x) & fp6_eq(a.y, b.y));
}
static inline proj jac_neg(proj s) { /* src/curve/jacobian.rs:443-449 */
    proj r = {s.x, fp6_neg(s.y), s.z};
    return r;
}

/* JacobianPoint::add, src/curve/jacobian.rs:452-505: the chord formula, then the four selections in the
 * reference's order (self identity -> rhs; rhs identity -> self; rhs == -self -> identity; rhs == self -> double) */
static proj jac_add(proj s, proj r) {
    fp6 z1_sq = fp6_square(s.z);
    fp6 z2_sq = fp6_square(r.z);
    fp6 a = fp6_mul(s.x, z2_sq);
    fp6 b = fp6_mul(r.x, z1_sq);
    fp6 c = fp6_mul(r.z, z2_sq);
    c = fp6_mul(c, s.y);
    fp6 d = fp6_mul(s.z, z1_sq);
    d = fp6_mul(d, r.y);
    fp6 e = fp6_sub(b, a);
    fp6 f = fp6_sub(d, c);
    fp6 e_sq = fp6_square(e);
    fp6 e_sq_e = fp6_mul(e, e_sq);
    fp6 a_e_sq = fp6_mul(a, e_sq);
    fp6 x3 = fp6_square(f);
    x3 = fp6_sub(x3, e_sq_e);
    x3 = fp6_sub(x3, fp6_double(a_e_sq));
    fp6 y3 = fp6_sub(a_e_sq, x3);
    y3 = fp6_mul(y3, f);
    fp6 t3 = fp6_mul(c, e_sq_e);
    y3 = fp6_sub(y3, t3);
    fp6 z3 = fp6_mul(s.z, r.z);
    z3 = fp6_mul(z3, e);
    proj tmp = {x3, y3, z3};
    if (fp6_is_zero(s.z)) tmp = r;
    if (fp6_is_zero(r.z)) tmp = s;
    if (jac_eq(r, jac_neg(s))) tmp = jac_identity();
    if (jac_eq(r, s)) tmp = jac_double(s);
    return tmp;
}

/* JacobianPoint::add_mixed, src/curve/jacobian.rs:508-552.  Its two comparisons go through
 * AffinePoint::from(JacobianPoint) (one Fp6 inversion each, src/curve/affine.rs:86-101), restated as such. */
static proj jac_add_mixed(proj s, affine r) {
    proj tmp = jac_add_mixed_unchecked(s, r); /* identical sequence, including the rhs-identity selection */
    if (affine_eq(r, affine_from_jac(jac_neg(s)))) tmp = jac_identity();
    if (affine_eq(r, affine_from_jac(s))) tmp = jac_double(s);
    return tmp;
}

/* src/curve/jacobian.rs:559-595 */
static proj jac_add_mixed_unchecked(proj s, affine r) {
    fp6 z1_sq = fp6_square(s.z);
    fp6 b = fp6_mul(r.x, z1_sq);
    fp6 d = fp6_mul(s.z, z1_sq);
    d = fp6_mul(d, r.y);
    fp6 e = fp6_sub(b, s.x);
    fp6 f = fp6_sub(d, s.y);
    fp6 e_sq = fp6_square(e);
    fp6 e_sq_e = fp6_mul(e, e_sq);
    fp6 x_e_sq = fp6_mul(s.x, e_sq);
    fp6 x3 = fp6_square(f);
    x3 = fp6_sub(x3, e_sq_e);
    x3 = fp6_sub(x3, fp6_double(x_e_sq));
    fp6 y3 = fp6_sub(x_e_sq, x3);
    y3 = fp6_mul(y3, f);
    fp6 t3 = fp6_mul(s.y, e_sq_e);
    y3 = fp6_sub(y3, t3);
    fp6 z3 = fp6_mul(s.z, e);
    proj o = {x3, y3, z3};
    return r.inf ? s : o;
}

/* src/curve/jacobian.rs:1094-1128 */
static mjac mjac_double_unchecked(mjac s) {
    fp6 x_sq = fp6_square(s.x);
    fp6 y_sq = fp6_square(s.y);
    fp6 y_sq2 = fp6_square(y_sq);
    fp6 a = fp6_mul_by_u32(y_sq, 4);
    a = fp6_mul(a, s.x);
    fp6 b = fp6_triple(x_sq);
    b = fp6_add(b, s.z4);
    fp6 c = fp6_mul_by_u32(y_sq2, 8);
    fp6 x3 = fp6_square(b);
    x3 = fp6_sub(x3, fp6_double(a));
    fp6 y3 = fp6_sub(a, x3);
    y3 = fp6_mul(y3, b);
    y3 = fp6_sub(y3, c);
    fp6 z3 = fp6_double(s.y);
    z3 = fp6_mul(z3, s.z);
    fp6 z34 = fp6_double(c);
    z34 = fp6_mul(z34, s.z4);
    mjac o = {x3, y3, z3, z34};
    return o;
}

/* src/curve/jacobian.rs:1146-1186 */
static mjac mjac_add_mixed_unchecked(mjac s, affine r) {
    fp6 z1_sq = fp6_square(s.z);
    fp6 b = fp6_mul(r.x, z1_sq);
    fp6 d = fp6_mul(s.z, z1_sq);
    d = fp6_mul(d, r.y);
    fp6 e = fp6_sub(b, s.x);
    fp6 f = fp6_sub(d, s.y);
    fp6 e_sq = fp6_square(e);
    fp6 e_sq_e = fp6_mul(e, e_sq);
    fp6 x_e_sq = fp6_mul(s.x, e_sq);
    fp6 x3 = fp6_square(f);
    x3 = fp6_sub(x3, e_sq_e);
    x3 = fp6_sub(x3, fp6_double(x_e_sq));
    fp6 y3 = fp6_sub(x_e_sq, x3);
    y3 = fp6_mul(y3, f);
    fp6 t3 = fp6_mul(s.y, e_sq_e);
    y3 = fp6_sub(y3, t3);
    fp6 z3 = fp6_mul(s.z, e);
    fp6 z34 = fp6_square(z3);
    z34 = fp6_square(z34);
    mjac o = {x3, y3, z3, z34};
    return r.inf ? s : o;
}

void ora_proj_double(const u64 *p, u64 *o) { proj_store(o, proj_double(proj_load(p))); }
void ora_proj_double_unchecked(const u64 *p, u64 *o) { proj_store(o, proj_double_unchecked(proj_load(p))); }
void ora_proj_add(const u64 *p, const u64 *q, u64 *o) { proj_store(o, proj_add(proj_load(p), proj_load(q))); }
void ora_proj_add_mixed(const u64 *p, const u64 *q, uint8_t qi, u64 *o) {
    proj_store(o, proj_add_mixed(proj_load(p), affine_load(q, qi)));
}
void ora_proj_add_mixed_unchecked(const u64 *p, const u64 *q, uint8_t qi, u64 *o) {
    proj_store(o, proj_add_mixed_unchecked(proj_load(p), affine_load(q, qi)));
}
void ora_proj_to_affine(const u64 *p, u64 *xy, uint8_t *inf) { affine_store(xy, inf, affine_from_proj(proj_load(p))); }
int ora_proj_is_on_curve(const u64 *p) { return proj_is_on_curve(proj_load(p)); }
void ora_jac_double(const u64 *p, u64 *o) { proj_store(o, jac_double(proj_load(p))); }
void ora_jac_double_unchecked(const u64 *p, u64 *o) { proj_store(o, jac_double_unchecked(proj_load(p))); }
void ora_jac_add_mixed_unchecked(const u64 *p, const u64 *q, uint8_t qi, u64 *o) {
    proj_store(o, jac_add_mixed_unchecked(proj_load(p), affine_load(q, qi)));
}
void ora_jac_add(const u64 *p, const u64 *q, u64 *o) { proj_store(o, jac_add(proj_load(p), proj_load(q))); }
void ora_jac_add_mixed(const u64 *p, const u64 *q, uint8_t qi, u64 *o) {
    proj_store(o, jac_add_mixed(proj_load(p), affine_load(q, qi)));
}
int ora_jac_eq(const u64 *p, const u64 *q) { return jac_eq(proj_load(p), proj_load(q)); }
int ora_proj_eq(const u64 *p, const u64 *q) { return proj_eq(proj_load(p), proj_load(q)); }
void ora_jac_to_affine(const u64 *p, u64 *xy, uint8_t *inf) { affine_store(xy, inf, affine_from_jac(proj_load(p))); }
int ora_jac_is_on_curve(const u64 *p) { return jac_is_on_curve(proj_load(p)); }
static inline mjac mjac_load(const u64 *p) {
    mjac r = {fp6_load(p), fp6_load(p + 6), fp6_load(p + 12), fp6_load(p + 18)};
    return r;
}
static inline void mjac_store(u64 *o, mjac p) {
    fp6_store(o, p.x);
    fp6_store(o + 6, p.y);
    fp6_store(o + 12, p.z);
    fp6_store(o + 18, p.z4);
}
void ora_mjac_double_unchecked(const u64 *p, u64 *o) { mjac_store(o, mjac_double_unchecked(mjac_load(p))); }
void ora_mjac_add_mixed_unchecked(const u64 *p, const u64 *q, uint8_t qi, u64 *o) {
    mjac_store(o, mjac_add_mixed_unchecked(mjac_load(p), affine_load(q, qi)));
}
void ora_mjac_to_affine(const u64 *p, u64 *xy, uint8_t *inf) { affine_store(xy, inf, affine_from_mjac(mjac_load(p))); }

/* ------------------------------------------------------ batch_normalize ---- */

/* src/curve/projective.rs:897-930 on in-memory structs */
static void proj_batch_normalize(const proj *p, affine *q, size_t n) {
    fp6 acc = FP6_ONE;
    for (size_t i = 0; i < n; i++) {
        q[i].x = acc;
        acc = fp6_select(fp6_mul(acc, p[i].z), acc, fp6_is_zero(p[i].z));
    }
    fp6 inv = FP6_ZERO;
    fp6_invert(acc, &inv); /* unwrap(): acc is a product of non-zero z's (or one) */
    acc = inv;
    for (size_t k = n; k-- > 0;) {
        int skip = fp6_is_zero(p[k].z);
        fp6 tmp = fp6_mul(q[k].x, acc);
        acc = fp6_select(fp6_mul(acc, p[k].z), acc, skip);
        q[k].x = fp6_mul(p[k].x, tmp);
        q[k].y = fp6_mul(p[k].y, tmp);
        q[k].inf = 0;
        if (skip) q[k] = affine_identity();
    }
}

/* src/curve/jacobian.rs:886-920 */
static void jac_batch_normalize(const proj *p, affine *q, size_t n) {
    fp6 acc = FP6_ONE;
    for (size_t i = 0; i < n; i++) {
        q[i].x = acc;
        acc = fp6_select(fp6_mul(acc, p[i].z), acc, fp6_is_zero(p[i].z));
    }
    fp6 inv = FP6_ZERO;
    fp6_invert(acc, &inv);
    acc = inv;
    for (size_t k = n; k-- > 0;) {
        int skip = fp6_is_zero(p[k].z);
        fp6 tmp = fp6_mul(q[k].x, acc);
        fp6 tmp2 = fp6_square(tmp);
        acc = fp6_select(fp6_mul(acc, p[k].z), acc, skip);
        q[k].x = fp6_mul(p[k].x, tmp2);
        q[k].y = fp6_mul(fp6_mul(p[k].y, tmp2), tmp);
        q[k].inf = 0;
        if (skip) q[k] = affine_identity();
    }
}

static void batch_normalize_flat(const u64 *p, size_t n, u64 *out_xy, uint8_t *out_inf, int jac) {
    if (n == 0) return;
    proj *pp = (proj *)malloc(n * sizeof(proj));
    affine *qq = (affine *)malloc(n * sizeof(affine));
    for (size_t i = 0; i < n; i++) pp[i] = proj_load(p + 18 * i);
    if (jac)
        jac_batch_normalize(pp, qq, n);
    else
        proj_batch_normalize(pp, qq, n);
    for (size_t i = 0; i < n; i++) affine_store(out_xy + 12 * i, out_inf ? out_inf + i : NULL, qq[i]);
    free(pp);
    free(qq);
}
void ora_proj_batch_normalize(const u64 *p, size_t n, u64 *xy, uint8_t *inf) { batch_normalize_flat(p, n, xy, inf, 0); }
void ora_jac_batch_normalize(const u64 *p, size_t n, u64 *xy, uint8_t *inf) { batch_normalize_flat(p, n, xy, inf, 1); }

/* -------------------------------------------------------------- tables ---- */

#define MAXTAB 128

/* src/lookup.rs:25-37 */
static void lookup_table_from_proj(proj p, int n, affine *out) {
    proj pts[MAXTAB];
    for (int i = 0; i < n; i++) pts[i] = p;
    for (int i = 1; i < n; i++) pts[i] = proj_add(p, pts[i - 1]);
    for (int i = 0; i < n; i++) out[i] = affine_identity();
    proj_batch_normalize(pts, out, (size_t)n);
}

/* src/naf_lookup.rs:35-48 */
static void naf_table_from_proj(proj p, int n, affine *out) {
    proj pts[MAXTAB];
    for (int i = 0; i < n; i++) pts[i] = p;
    proj dbl = proj_double(p);
    for (int i = 1; i < n; i++) pts[i] = proj_add(dbl, pts[i - 1]);
    for (int i = 0; i < n; i++) out[i] = affine_identity();
    proj_batch_normalize(pts, out, (size_t)n);
}

/* src/lookup.rs:81-102 (constant-time scan restated as the value it selects) */
static affine lookup_get_point(const affine *tab, int n, int8_t x) {
    int16_t xmask = (int16_t)((int16_t)x >> 7);
    int16_t xabs = (int16_t)(((int16_t)x + xmask) ^ xmask);
    affine t = affine_identity();
    for (int j = 1; j < n + 1; j++) {
        if ((uint8_t)xabs == (uint8_t)j) t = tab[j - 1];
    }
    if (xmask & 1) t = affine_neg(t);
    return t;
}

void ora_lookup_table_from_proj(const u64 *p, int n, u64 *out_xy, uint8_t *out_inf) {
    affine t[MAXTAB];
    lookup_table_from_proj(proj_load(p), n, t);
    for (int i = 0; i < n; i++) affine_store(out_xy + 12 * i, out_inf ? out_inf + i : NULL, t[i]);
}
void ora_naf_table_from_proj(const u64 *p, int n, u64 *out_xy, uint8_t *out_inf) {
    affine t[MAXTAB];
    naf_table_from_proj(proj_load(p), n, t);
    for (int i = 0; i < n; i++) affine_store(out_xy + 12 * i, out_inf ? out_inf + i : NULL, t[i]);
}

/* src/curve/affine.rs:255-275 */
static const affine GENERATOR = {
    {{0x263a588f4b0118a1ULL, 0x7757a0bcb26a142dULL, 0x9215adfc1e925890ULL, 0x430aad2ce14759a4ULL, 0x534ece54de4b2c8ULL,
      0xb39050f01f7b1f33ULL}},
    {{0xd57f0d0d47482534ULL, 0x26821d894fa8ea0fULL, 0xc77f564783ef13a1ULL, 0x949c360784284ec2ULL, 0xb7040bd639ef3cc4ULL,
      0x8aa635f2719d255fULL}},
    0};
/* src/constants.rs:43-61 */
static const affine SHIFT_POINT = {
    {{0xab95fb0d188ff2b1ULL, 0xf38263a8c954493eULL, 0x1bc968d6739028b7ULL, 0x4f67684fe6b27d1fULL, 0x6504f8bbba9077b4ULL,
      0x860838116d76e545ULL}},
    {{0x56988918d5b02cf3ULL, 0xc9fe3ef9d7652c54ULL, 0x4923e90c4d6f4538ULL, 0xc747ac3f82369f27ULL, 0x6e74c106972c71f5ULL,
      0xcf190938f1c64bb5ULL}},
    0};

static affine G_MINUS_SHIFT[257];    /* src/constants.rs:76-90 */
static affine G_BASEPOINT_TABLE[32][8]; /* src/constants.rs:21-22, src/lookup.rs:164-173 */
static affine G_BASEPOINT_LOOKUP[128]; /* src/constants.rs:26-27 */
static u64 G_FLAT_MINUS_SHIFT[257 * 12], G_FLAT_BPT[32 * 8 * 12], G_FLAT_BPL[128 * 12], G_FLAT_SHIFT[12], G_FLAT_GEN[12];
static affine G_ODD_MULTIPLES[64];   /* ODD_MULTIPLES_BASEPOINT: NafLookupTable<64> of G, src/constants.rs:32-33 */
static u64 G_FLAT_ODD[64 * 12];
static int g_init_done = 0, g_bpl_done = 0, g_odd_done = 0;

static void ora_init(void) {
    if (g_init_done) return;
#pragma omp critical(ora_init_lock)
    {
        if (!g_init_done) {
            /* generate_opposite_powers, src/constants.rs:80-90 */
            static proj arr[257];
            proj sp = proj_from_affine(SHIFT_POINT);
            proj nsp = {sp.x, fp6_neg(sp.y), sp.z};
            arr[0] = nsp;
            for (int i = 1; i < 257; i++) arr[i] = proj_double(arr[i - 1]);
            proj_batch_normalize(arr, G_MINUS_SHIFT, 257);
            /* BasePointTable::create, src/lookup.rs:164-173 */
            proj point = proj_from_affine(GENERATOR);
            lookup_table_from_proj(point, 8, G_BASEPOINT_TABLE[0]);
            for (int i = 1; i < 32; i++) {
                for (int k = 0; k < 8; k++) point = proj_double(point); /* double_multi(8) */
                lookup_table_from_proj(point, 8, G_BASEPOINT_TABLE[i]);
            }
            for (int i = 0; i < 257; i++) affine_store(G_FLAT_MINUS_SHIFT + 12 * i, NULL, G_MINUS_SHIFT[i]);
            for (int i = 0; i < 32; i++)
                for (int j = 0; j < 8; j++) affine_store(G_FLAT_BPT + 12 * (8 * i + j), NULL, G_BASEPOINT_TABLE[i][j]);
            affine_store(G_FLAT_SHIFT, NULL, SHIFT_POINT);
            affine_store(G_FLAT_GEN, NULL, GENERATOR);
            __sync_synchronize();
            g_init_done = 1;
        }
    }
}
static void ora_init_bpl(void) {
    if (g_bpl_done) return;
#pragma omp critical(ora_init_bpl_lock)
    {
        if (!g_bpl_done) {
            lookup_table_from_proj(proj_from_affine(GENERATOR), 128, G_BASEPOINT_LOOKUP);
            for (int i = 0; i < 128; i++) affine_store(G_FLAT_BPL + 12 * i, NULL, G_BASEPOINT_LOOKUP[i]);
            __sync_synchronize();
            g_bpl_done = 1;
        }
    }
}
static void ora_init_odd(void) {
    if (g_odd_done) return;
#pragma omp critical(ora_init_odd_lock)
    {
        if (!g_odd_done) {
            naf_table_from_proj(proj_from_affine(GENERATOR), 64, G_ODD_MULTIPLES);
            for (int i = 0; i < 64; i++) affine_store(G_FLAT_ODD + 12 * i, NULL, G_ODD_MULTIPLES[i]);
            __sync_synchronize();
            g_odd_done = 1;
        }
    }
}
const u64 *ora_odd_multiples_basepoint(void) {
    ora_init_odd();
    return G_FLAT_ODD;
}
const u64 *ora_basepoint_table(void) {
    ora_init();
    return G_FLAT_BPT;
}
const u64 *ora_basepoint_lookup(void) {
    ora_init_bpl();
    return G_FLAT_BPL;
}
const u64 *ora_minus_shift_point_array(void) {
    ora_init();
    return G_FLAT_MINUS_SHIFT;
}
const u64 *ora_shift_point_affine(void) {
    ora_init();
    return G_FLAT_SHIFT;
}
const u64 *ora_generator_affine(void) {
    ora_init();
    return G_FLAT_GEN;
}

/* ------------------------------------------------------------- ladders ---- */

/* src/curve/projective.rs:614-625 */
static proj proj_multiply(proj self, const uint8_t *by) {
    ora_init();
    affine table[16];
    lookup_table_from_proj(self, 16, table);
    int8_t digits[64];
    ora_bytes_to_radix_16(by, digits);
    proj acc = proj_from_affine(SHIFT_POINT);
    for (int i = 63; i >= 0; i--) {
        for (int k = 0; k < 4; k++) acc = proj_double_unchecked(acc);
        acc = proj_add_mixed_unchecked(acc, lookup_get_point(table, 16, digits[i]));
    }
    return proj_add_mixed_unchecked(acc, G_MINUS_SHIFT[256]);
}

/* src/curve/projective.rs:633-662, including quirk Q1 (i stays 255) */
static proj proj_multiply_vartime(proj self, const uint8_t *by) {
    ora_init();
    int8_t digits[256];
    ora_bytes_to_wnaf_vartime(by, 5, digits);
    size_t i = 255;
    for (size_t j = 256; j-- > 0;) {
        if (digits[i] != 0) {
            i = j;
            break;
        }
    }
    affine table[8];
    naf_table_from_proj(self, 8, table);
    proj acc = proj_from_affine(SHIFT_POINT);
    for (size_t j = i + 1; j-- > 0;) {
        acc = proj_double_unchecked(acc);
        if (digits[j] > 0)
            acc = proj_add_mixed_unchecked(acc, table[(size_t)digits[j] / 2]);
        else if (digits[j] < 0)
            acc = proj_add_mixed_unchecked(acc, affine_neg(table[(size_t)(-digits[j]) / 2]));
    }
    return proj_add_mixed_unchecked(acc, G_MINUS_SHIFT[i + 1]);
}

/* src/lookup.rs:189-206 */
static proj basepoint_table_multiply(const uint8_t *by) {
    ora_init();
    int8_t a[64];
    ora_bytes_to_radix_16(by, a);
    proj acc = proj_from_affine(SHIFT_POINT);
    for (int i = 1; i < 64; i += 2) acc = proj_add_mixed_unchecked(acc, lookup_get_point(G_BASEPOINT_TABLE[i / 2], 8, a[i]));
    for (int k = 0; k < 4; k++) acc = proj_double_unchecked(acc);
    for (int i = 0; i < 64; i += 2) acc = proj_add_mixed_unchecked(acc, lookup_get_point(G_BASEPOINT_TABLE[i / 2], 8, a[i]));
    return proj_add_mixed_unchecked(acc, G_MINUS_SHIFT[4]);
}

static inline proj jac_from_mjac(mjac p) { /* src/curve/jacobian.rs:118-126 */
    proj r = {p.x, p.y, p.z};
    return r;
}

/* src/constants.rs:124-135 */
static proj basepoint_lookup_multiply(const uint8_t *by) {
    ora_init();
    ora_init_bpl();
    int8_t digits[32];
    ora_bytes_to_radix_256(by, digits);
    mjac acc = mjac_from_affine(SHIFT_POINT);
    for (int i = 31; i >= 0; i--) {
        for (int k = 0; k < 8; k++) acc = mjac_double_unchecked(acc);
        acc = mjac_add_mixed_unchecked(acc, lookup_get_point(G_BASEPOINT_LOOKUP, 128, digits[i]));
    }
    return jac_from_mjac(mjac_add_mixed_unchecked(acc, G_MINUS_SHIFT[256]));
}

/* src/curve/jacobian.rs:599-611; LookupTable::<16>::from(&JacobianPoint) goes
 * through AffinePoint::from(p) then ProjectivePoint::from (src/lookup.rs:45-61) */
static proj jac_multiply(proj self, const uint8_t *by) {
    ora_init();
    affine table[16];
    lookup_table_from_proj(proj_from_affine(affine_from_jac(self)), 16, table);
    int8_t digits[64];
    ora_bytes_to_radix_16(by, digits);
    mjac acc = mjac_from_affine(SHIFT_POINT);
    for (int i = 63; i >= 0; i--) {
        for (int k = 0; k < 4; k++) acc = mjac_double_unchecked(acc);
        acc = mjac_add_mixed_unchecked(acc, lookup_get_point(table, 16, digits[i]));
    }
    return jac_from_mjac(mjac_add_mixed_unchecked(acc, G_MINUS_SHIFT[256]));
}

/* src/curve/jacobian.rs:619-649 (Q1 applies) */
static proj jac_multiply_vartime(proj self, const uint8_t *by) {
    ora_init();
    int8_t digits[256];
    ora_bytes_to_wnaf_vartime(by, 5, digits);
    size_t i = 255;
    for (size_t j = 256; j-- > 0;) {
        if (digits[i] != 0) {
            i = j;
            break;
        }
    }
    affine table[8];
    naf_table_from_proj(proj_from_affine(affine_from_jac(self)), 8, table);
    mjac acc = mjac_from_affine(SHIFT_POINT);
    for (size_t j = i + 1; j-- > 0;) {
        acc = mjac_double_unchecked(acc);
        if (digits[j] > 0)
            acc = mjac_add_mixed_unchecked(acc, table[(size_t)digits[j] / 2]);
        else if (digits[j] < 0)
            acc = mjac_add_mixed_unchecked(acc, affine_neg(table[(size_t)(-digits[j]) / 2]));
    }
    return jac_from_mjac(mjac_add_mixed_unchecked(acc, G_MINUS_SHIFT[i + 1]));
}

void ora_proj_multiply(const u64 *p, const uint8_t *by, u64 *o) { proj_store(o, proj_multiply(proj_load(p), by)); }
void ora_proj_multiply_vartime(const u64 *p, const uint8_t *by, u64 *o) {
    proj_store(o, proj_multiply_vartime(proj_load(p), by));
}
void ora_basepoint_table_multiply(const uint8_t *by, u64 *o) { proj_store(o, basepoint_table_multiply(by)); }
void ora_basepoint_lookup_multiply(const uint8_t *by, u64 *o) { proj_store(o, basepoint_lookup_multiply(by)); }
void ora_jac_multiply(const u64 *p, const uint8_t *by, u64 *o) { proj_store(o, jac_multiply(proj_load(p), by)); }
void ora_jac_multiply_vartime(const u64 *p, const uint8_t *by, u64 *o) {
    proj_store(o, jac_multiply_vartime(proj_load(p), by));
}

/* -------------------------------------------------------- batch drivers ---- */

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ora_fp6_mul_batch(const u64 *a, const u64 *b, u64 *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) fp6_store(out + 6 * i, fp6_canonical(fp6_mul(fp6_load(a + 6 * i), fp6_load(b + 6 * i))));
}
void ora_fp6_square_batch(const u64 *a, u64 *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) fp6_store(out + 6 * i, fp6_canonical(fp6_square(fp6_load(a + 6 * i))));
}
void ora_fp6_invert_batch(const u64 *a, u64 *out, uint8_t *is_some, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        fp6 r = FP6_ZERO;
        int s = fp6_invert(fp6_load(a + 6 * i), &r);
        fp6_store(out + 6 * i, s ? fp6_canonical(r) : FP6_ZERO);
        if (is_some) is_some[i] = (uint8_t)s;
    }
}

/* in_form / out_form: 0 = affine, 1 = projective, 2 = Jacobian.  Which reference
 * method a pair names: PROJECTIVE on either side -> ProjectivePoint::multiply*,
 * otherwise JacobianPoint::multiply* (AffinePoint::multiply* delegates to it,
 * src/curve/affine.rs:373-386). */
static void emit_point(proj r, int jac, u64 *out, uint8_t *out_inf, size_t i, int out_form) {
    if (out_form == 0) {
        affine a = jac ? affine_from_jac(r) : affine_from_proj(r); /* src/curve/affine.rs:64-101 */
        affine_store(out + 12 * i, out_inf ? out_inf + i : NULL, a);
    } else {
        proj c = {fp6_canonical(r.x), fp6_canonical(r.y), fp6_canonical(r.z)};
        proj_store(out + 18 * i, c);
        if (out_inf) out_inf[i] = (uint8_t)fp6_is_zero(r.z);
    }
}

static proj load_in_point(const u64 *pts, int in_form, size_t i) {
    if (in_form == 0) { /* affine x|y; (0, 1) is taken as the identity encoding, src/curve/affine.rs:224-230 */
        affine a = affine_load(pts + 12 * i, 0);
        a.inf = (uint8_t)(fp6_is_zero(a.x) && fp6_eq(a.y, FP6_ONE));
        return proj_from_affine(a); /* JacobianPoint::from(&AffinePoint) has the same coordinates, src/curve/jacobian.rs:102-110 */
    }
    return proj_load(pts + 18 * i);
}

void ora_mul_by_generator_batch(const uint8_t *scalars, u64 *out, uint8_t *out_inf, size_t n, int out_form) {
    ora_init();
    if (out_form == 2) ora_init_bpl();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        if (out_form == 2) /* &BASEPOINT_LOOKUP * &Scalar, src/constants.rs:102-117 */
            emit_point(basepoint_lookup_multiply(scalars + 32 * i), 1, out, out_inf, i, out_form);
        else
            emit_point(basepoint_table_multiply(scalars + 32 * i), 0, out, out_inf, i, out_form);
    }
}
/* BasePointTable::create(&P) for any base (src/lookup.rs:164-173): 32 x LookupTable<8> of 256^i P, flat, table-major */
void ora_basepoint_table_create(const u64 *p_in, int in_form, u64 *out_xy, uint8_t *out_inf) {
    proj point = load_in_point(p_in, in_form, 0);
    affine t[8];
    for (int i = 0; i < 32; i++) {
        if (i > 0)
            for (int k = 0; k < 8; k++) point = proj_double(point); /* double_multi(8) */
        lookup_table_from_proj(point, 8, t);
        for (int j = 0; j < 8; j++) affine_store(out_xy + 12 * (8 * i + j), out_inf + 8 * i + j, t[j]);
    }
}
/* BasePointTable::multiply with a caller's table (src/lookup.rs:189-206), over a batch */
void ora_basepoint_table_multiply_batch(const u64 *table_xy, const uint8_t *table_inf, const uint8_t *scalars, u64 *out,
                                        uint8_t *out_inf, size_t n, int out_form) {
    ora_init();
    affine tab[32][8];
    for (int i = 0; i < 256; i++) tab[i / 8][i % 8] = affine_load(table_xy + 12 * i, table_inf[i]);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        int8_t a[64];
        ora_bytes_to_radix_16(scalars + 32 * i, a);
        proj acc = proj_from_affine(SHIFT_POINT);
        for (int j = 1; j < 64; j += 2) acc = proj_add_mixed_unchecked(acc, lookup_get_point(tab[j / 2], 8, a[j]));
        for (int k = 0; k < 4; k++) acc = proj_double_unchecked(acc);
        for (int j = 0; j < 64; j += 2) acc = proj_add_mixed_unchecked(acc, lookup_get_point(tab[j / 2], 8, a[j]));
        emit_point(proj_add_mixed_unchecked(acc, G_MINUS_SHIFT[4]), 0, out, out_inf, i, out_form);
    }
}
/* out[i] = lhs[i] + rhs[i] with the complete formulas.  lhs_form 1: ProjectivePoint::add (src/curve/projective.rs:456-515) or,
 * rhs_form 0, ::add_mixed (518-569); lhs_form 2: JacobianPoint::add (src/curve/jacobian.rs:452-505) or ::add_mixed (508-552).
 * out: the reference's own representative in lhs_form, canonical limbs. */
void ora_add_batch(const u64 *lhs, int lhs_form, const u64 *rhs, int rhs_form, u64 *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        proj s = proj_load(lhs + 18 * i), o;
        if (rhs_form == 0) {
            affine r = affine_load(rhs + 12 * i, 0);
            r.inf = (uint8_t)(fp6_is_zero(r.x) && fp6_eq(r.y, FP6_ONE));
            o = lhs_form == 1 ? proj_add_mixed(s, r) : jac_add_mixed(s, r);
        } else {
            proj r = proj_load(rhs + 18 * i);
            o = lhs_form == 1 ? proj_add(s, r) : jac_add(s, r);
        }
        emit_point(o, 0, out, NULL, i, 1);
    }
}
/* ProjectivePoint::double (src/curve/projective.rs:349-393) / JacobianPoint::double (src/curve/jacobian.rs:368-402) */
void ora_double_batch(const u64 *pts, int form, u64 *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        proj s = proj_load(pts + 18 * i);
        emit_point(form == 1 ? proj_double(s) : jac_double(s), 0, out, NULL, i, 1);
    }
}
/* ProjectivePoint / JacobianPoint::double_multi(k): k >= 1 iterated ::double (src/curve/projective.rs:322-331,
 * src/curve/jacobian.rs:341-350) */
void ora_double_multi_batch(const u64 *pts, int form, uint32_t k, u64 *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        proj s = proj_load(pts + 18 * i);
        for (uint32_t r = 0; r < k; r++) s = form == 1 ? proj_double(s) : jac_double(s);
        emit_point(s, 0, out, NULL, i, 1);
    }
}
/* The incomplete ("unchecked") forms: ::double_multi_unchecked(k) = k iterated ::double_unchecked (src/curve/projective.rs:337-346,
 * :400-443; src/curve/jacobian.rs:356-365, :409-439) and ::add_mixed_unchecked (src/curve/projective.rs:576-610,
 * src/curve/jacobian.rs:559-595) */
void ora_double_multi_unchecked_batch(const u64 *pts, int form, uint32_t k, u64 *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        proj s = proj_load(pts + 18 * i);
        for (uint32_t r = 0; r < k; r++) s = form == 1 ? proj_double_unchecked(s) : jac_double_unchecked(s);
        emit_point(s, 0, out, NULL, i, 1);
    }
}
void ora_add_mixed_unchecked_batch(const u64 *lhs, int form, const u64 *rhs_xy, u64 *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        proj s = proj_load(lhs + 18 * i);
        affine r = affine_load(rhs_xy + 12 * i, 0);
        r.inf = (uint8_t)(fp6_is_zero(r.x) && fp6_eq(r.y, FP6_ONE));
        emit_point(form == 1 ? proj_add_mixed_unchecked(s, r) : jac_add_mixed_unchecked(s, r), 0, out, NULL, i, 1);
    }
}
/* -P: AffinePoint::neg (src/curve/affine.rs:164-175: the identity keeps y = 1), ProjectivePoint::neg
 * (src/curve/projective.rs:447-453), JacobianPoint::neg (src/curve/jacobian.rs:443-449); canonical limbs */
void ora_neg_batch(const u64 *pts, int form, u64 *out, size_t n) {
    const size_t w = form == 0 ? 12 : 18;
    for (size_t i = 0; i < n; i++) {
        fp6 x = fp6_load(pts + w * i), y = fp6_load(pts + w * i + 6);
        if (form == 0) {
            affine a = {x, y, (uint8_t)(fp6_is_zero(x) && fp6_eq(y, FP6_ONE))};
            a = affine_neg(a);
            x = a.x;
            y = a.y;
        } else {
            y = fp6_neg(y);
            fp6_store(out + w * i + 12, fp6_canonical(fp6_load(pts + w * i + 12)));
        }
        fp6_store(out + w * i, fp6_canonical(x));
        fp6_store(out + w * i + 6, fp6_canonical(y));
    }
}
/* flags[i] = a[i].ct_eq(&b[i]): ConstantTimeEq for AffinePoint (src/curve/affine.rs:132-144, (0, 1) = the identity at the ABI),
 * ProjectivePoint (src/curve/projective.rs:118-135), JacobianPoint (src/curve/jacobian.rs:134-154) */
void ora_eq_batch(const u64 *a, const u64 *b, int form, uint8_t *flags, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (form == 0) {
            affine p = affine_load(a + 12 * i, 0), q = affine_load(b + 12 * i, 0);
            p.inf = (uint8_t)(fp6_is_zero(p.x) && fp6_eq(p.y, FP6_ONE));
            q.inf = (uint8_t)(fp6_is_zero(q.x) && fp6_eq(q.y, FP6_ONE));
            flags[i] = (uint8_t)affine_eq(p, q);
        } else {
            proj p = proj_load(a + 18 * i), q = proj_load(b + 18 * i);
            flags[i] = (uint8_t)(form == 1 ? proj_eq(p, q) : jac_eq(p, q));
        }
    }
}
/* AffinePoint::is_on_curve, src/curve/affine.rs:364-368; (0, 1) is the identity's encoding at the ABI */
int ora_affine_is_on_curve(const u64 *xy) {
    affine a = affine_load(xy, 0);
    if (fp6_is_zero(a.x) && fp6_eq(a.y, FP6_ONE)) return 1;
    fp6 lhs = fp6_sub(fp6_square(a.y), fp6_mul(a.x, fp6_add(fp6_square(a.x), FP6_ONE)));
    return fp6_eq(lhs, CURVE_B);
}
void ora_multiply_batch(const u64 *pts, int in_form, const uint8_t *scalars, u64 *out, uint8_t *out_inf, size_t n,
                        int out_form) {
    ora_init();
    const int jac = !(in_form == 1 || out_form == 1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        proj p = load_in_point(pts, in_form, i);
        emit_point(jac ? jac_multiply(p, scalars + 32 * i) : proj_multiply(p, scalars + 32 * i), jac, out, out_inf, i, out_form);
    }
}
void ora_multiply_vartime_batch(const u64 *pts, int in_form, const uint8_t *scalars, u64 *out, uint8_t *out_inf, size_t n,
                                int out_form) {
    ora_init();
    const int jac = !(in_form == 1 || out_form == 1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        proj p = load_in_point(pts, in_form, i);
        emit_point(jac ? jac_multiply_vartime(p, scalars + 32 * i) : proj_multiply_vartime(p, scalars + 32 * i), jac, out, out_inf,
                   i, out_form);
    }
}


/* ---- multi-scalar ladders (Straus: shared doublings, one table per point) ---- */

#define MAXMANY 64

/* ProjectivePoint::multiply_many, src/curve/projective.rs:817-832; multiply_double
 * (:666-685) is the same sequence with two points. */
static proj proj_multiply_many(const proj *pts, const uint8_t *scalars, int m) {
    ora_init();
    static _Thread_local affine tables[MAXMANY][16];
    static _Thread_local int8_t digits[MAXMANY][64];
    for (int j = 0; j < m; j++) {
        lookup_table_from_proj(pts[j], 16, tables[j]);
        ora_bytes_to_radix_16(scalars + 32 * j, digits[j]);
    }
    proj acc = proj_from_affine(SHIFT_POINT);
    for (int i = 63; i >= 0; i--) {
        for (int k = 0; k < 4; k++) acc = proj_double_unchecked(acc);
        for (int j = 0; j < m; j++) acc = proj_add_mixed_unchecked(acc, lookup_get_point(tables[j], 16, digits[j][i]));
    }
    return proj_add_mixed_unchecked(acc, G_MINUS_SHIFT[256]);
}
/* JacobianPoint::multiply_many, src/curve/jacobian.rs:807-823; multiply_double :653-673 */
static proj jac_multiply_many(const proj *pts, const uint8_t *scalars, int m) {
    ora_init();
    static _Thread_local affine tables[MAXMANY][16];
    static _Thread_local int8_t digits[MAXMANY][64];
    for (int j = 0; j < m; j++) {
        lookup_table_from_proj(proj_from_affine(affine_from_jac(pts[j])), 16, tables[j]);
        ora_bytes_to_radix_16(scalars + 32 * j, digits[j]);
    }
    mjac acc = mjac_from_affine(SHIFT_POINT);
    for (int i = 63; i >= 0; i--) {
        for (int k = 0; k < 4; k++) acc = mjac_double_unchecked(acc);
        for (int j = 0; j < m; j++) acc = mjac_add_mixed_unchecked(acc, lookup_get_point(tables[j], 16, digits[j][i]));
    }
    return jac_from_mjac(mjac_add_mixed_unchecked(acc, G_MINUS_SHIFT[256]));
}

/* The vartime multi-scalar ladders.  multiply_many_vartime always runs 256 doublings
 * (src/curve/projective.rs:840-869, src/curve/jacobian.rs:831-858); multiply_double_vartime
 * and multiply_double_with_basepoint_vartime carry quirk Q1 (their "skip unset digits"
 * loop tests index i, which stays 255: src/curve/projective.rs:703-709, :765-771), so they
 * too run 256 doublings and correct with MINUS_SHIFT_POINT_ARRAY[256].
 * with_basepoint: the last scalar multiplies G through ODD_MULTIPLES_BASEPOINT with a
 * width-8 NAF (src/curve/projective.rs:756-809). */
static affine naf_pick(const affine *table, int8_t d) { /* NafLookupTable::get_point(x) = table[x / 2], src/naf_lookup.rs:87-92 */
    return d > 0 ? table[(size_t)d / 2] : affine_neg(table[(size_t)(-d) / 2]);
}
static proj multiply_many_vartime_generic(const proj *pts, const uint8_t *scalars, int m, int with_basepoint, int jac) {
    ora_init();
    if (with_basepoint) ora_init_odd();
    static _Thread_local affine tables[MAXMANY][8];
    static _Thread_local int8_t digits[MAXMANY + 1][256];
    for (int j = 0; j < m; j++) {
        naf_table_from_proj(jac ? proj_from_affine(affine_from_jac(pts[j])) : pts[j], 8, tables[j]);
        ora_bytes_to_wnaf_vartime(scalars + 32 * j, 5, digits[j]);
    }
    if (with_basepoint) ora_bytes_to_wnaf_vartime(scalars + 32 * m, 8, digits[m]);
    proj pacc = proj_from_affine(SHIFT_POINT);
    mjac macc = mjac_from_affine(SHIFT_POINT);
    for (int i = 255; i >= 0; i--) {
        if (jac) macc = mjac_double_unchecked(macc); else pacc = proj_double_unchecked(pacc);
        for (int j = 0; j < m + (with_basepoint ? 1 : 0); j++) {
            int8_t d = digits[j][i];
            if (d == 0) continue;
            affine q = j < m ? naf_pick(tables[j], d) : naf_pick(G_ODD_MULTIPLES, d);
            if (jac) macc = mjac_add_mixed_unchecked(macc, q); else pacc = proj_add_mixed_unchecked(pacc, q);
        }
    }
    if (jac) return jac_from_mjac(mjac_add_mixed_unchecked(macc, G_MINUS_SHIFT[256]));
    return proj_add_mixed_unchecked(pacc, G_MINUS_SHIFT[256]);
}

/* Batch driver: group g uses points / scalars [g*m, (g+1)*m) (with_basepoint: m points and
 * m + 1 scalars per group, the last one for G).  Forms as in ora_multiply_batch. */
void ora_multiply_many_batch(const u64 *pts, int in_form, const uint8_t *scalars, int m, int vartime, int with_basepoint,
                             u64 *out, uint8_t *out_inf, size_t n_groups, int out_form) {
    ora_init();
    const int jac = !(in_form == 1 || out_form == 1);
    const int ns = m + (with_basepoint ? 1 : 0);
#pragma omp parallel for schedule(static)
    for (size_t g = 0; g < n_groups; g++) {
        proj p[MAXMANY];
        for (int j = 0; j < m; j++) p[j] = load_in_point(pts, in_form, g * (size_t)m + j);
        const uint8_t *sc = scalars + 32 * g * (size_t)ns;
        proj r;
        if (vartime)
            r = multiply_many_vartime_generic(p, sc, m, with_basepoint, jac);
        else
            r = jac ? jac_multiply_many(p, sc, m) : proj_multiply_many(p, sc, m);
        emit_point(r, jac, out, out_inf, g, out_form);
    }
}

/* ------------------------------------------------- Fp3, square roots ---- */
/* Fp3 = Fp[X]/(X^3 - 7), embedded in Fp6 as (c0, c2, c4) (src/fp3.rs:41-48, :112-120). */
typedef struct { u64 a[3]; } fp3;
static const fp3 FP3_ZERO = {{0, 0, 0}};
static const fp3 FP3_ONE = {{1, 0, 0}};
#define OFF8P ((u128)0x7fffffff8ULL << 32 | 0x00000008ULL) /* 0x7fffffff800000008 = 8p, src/fp3.rs:186 */
#define OFF8P7 (OFF8P * 7)                                  /* 0x37ffffffc800000038 */

/* The cubic Karatsuba block shared by Fp3::mul (src/fp3.rs:169-211) and
 * Fp6::mul_by_{low,high}_fp3 (src/fp6.rs:469-647): unreduced d0, d1, d2. */
static inline void cubic_karatsuba(const u64 x[3], const u64 y[3], u128 *d0, u128 *d1, u128 *d2) {
    u128 t0 = fp_mul(x[0], y[0]), t1 = fp_mul(x[1], y[1]), t2 = fp_mul(x[2], y[2]);
    u128 s12 = fp_mul(fp_add(x[1], x[2]), fp_add(y[1], y[2]));
    u128 s01 = fp_mul(fp_add(x[0], x[1]), fp_add(y[0], y[1]));
    u128 s02 = fp_mul(fp_add(x[0], x[2]), fp_add(y[0], y[2]));
    *d0 = (s12 + OFF8P - (t1 + t2)) * 7 + t0;
    *d1 = t2 * 7 + OFF8P - (t0 + t1) + s01;
    *d2 = t1 + OFF8P - (t0 + t2) + s02;
}
static fp3 fp3_mul(fp3 x, fp3 y) { /* src/fp3.rs:169-211 */
    u128 d0, d1, d2;
    cubic_karatsuba(x.a, y.a, &d0, &d1, &d2);
    fp3 r = {{reduce_u96(d0), reduce_u96(d1), reduce_u96(d2)}};
    return r;
}
static fp3 fp3_square(fp3 x) { return fp3_mul(x, x); } /* src/fp3.rs:215-252 (same values) */
static int fp3_eq(fp3 x, fp3 y) {
    return fp_canonical(x.a[0]) == fp_canonical(y.a[0]) && fp_canonical(x.a[1]) == fp_canonical(y.a[1]) &&
           fp_canonical(x.a[2]) == fp_canonical(y.a[2]);
}
static fp3 fp3_exp_vartime(fp3 x, const u64 *power, int nwords) { /* src/fp3.rs:285-297 */
    fp3 res = FP3_ONE;
    for (int w = nwords - 1; w >= 0; w--)
        for (int i = 63; i >= 0; i--) {
            res = fp3_square(res);
            if ((power[w] >> i) & 1) res = fp3_mul(res, x);
        }
    return res;
}
/* Fp3::sqrt, Tonelli-Shanks for 2-adicity 32, src/fp3.rs:255-282 */
static int fp3_sqrt(fp3 x, fp3 *out) {
    static const u64 E[3] = {0x80000002fffffffeULL, 0x80000002fffffffcULL, 0x000000007ffffffeULL};
    fp3 y = fp3_exp_vartime(x, E, 3);
    fp3 s = fp3_mul(x, y), t = fp3_mul(s, y);
    fp3 z = {{2800184025912956819ULL, 0, 0}}; /* TWO_ADIC_ROOT_OF_UNITY_P3, src/fp3.rs:33-37 */
    for (int k = 32; k >= 2; k--) {
        fp3 b = t;
        for (int i = 0; i < k - 2; i++) b = fp3_square(b);
        int is_one = fp3_eq(b, FP3_ONE);
        fp3 new_s = fp3_mul(s, z);
        if (!is_one) s = new_s;
        z = fp3_square(z);
        fp3 new_t = fp3_mul(t, z);
        if (!is_one) t = new_t;
    }
    *out = s;
    return fp3_eq(fp3_square(s), x);
}
int ora_fp3_mul(const u64 *a, const u64 *b, u64 *o) {
    fp3 x = {{a[0], a[1], a[2]}}, y = {{b[0], b[1], b[2]}}, r = fp3_mul(x, y);
    for (int i = 0; i < 3; i++) o[i] = fp_canonical(r.a[i]);
    return 1;
}
int ora_fp3_sqrt(const u64 *a, u64 *o) {
    fp3 x = {{a[0], a[1], a[2]}}, r;
    int ok = fp3_sqrt(x, &r);
    for (int i = 0; i < 3; i++) o[i] = ok ? fp_canonical(r.a[i]) : 0;
    return ok;
}
/* Fp6::mul_by_low_fp3 / mul_by_high_fp3, src/fp6.rs:469-647: (A0 + A1 Y) * f and (A0 + A1 Y) * (f Y), Y^2 = X */
static fp6 fp6_mul_by_low_fp3(fp6 s, fp3 f) {
    u64 a0[3] = {s.c[0], s.c[2], s.c[4]}, a1[3] = {s.c[1], s.c[3], s.c[5]};
    u128 d00, d01, d02, d30, d31, d32;
    cubic_karatsuba(a0, f.a, &d00, &d01, &d02);
    cubic_karatsuba(a1, f.a, &d30, &d31, &d32);
    fp6 r = {{reduce_u96(d00 + OFF8P7), reduce_u96(d30 + OFF8P7), reduce_u96(d01 + OFF8P7), reduce_u96(d31 + OFF8P),
              reduce_u96(d02 + OFF8P), reduce_u96(d32 + OFF8P)}};
    return r;
}
static fp6 fp6_mul_by_high_fp3(fp6 s, fp3 f) {
    u64 a0[3] = {s.c[0], s.c[2], s.c[4]}, a1[3] = {s.c[1], s.c[3], s.c[5]};
    u128 d10, d11, d12, d20, d21, d22;
    cubic_karatsuba(a1, f.a, &d10, &d11, &d12);
    cubic_karatsuba(a0, f.a, &d20, &d21, &d22);
    fp6 r = {{reduce_u96((d12 + OFF8P) * 7), reduce_u96(d20 + OFF8P7), reduce_u96(d10 + OFF8P), reduce_u96(d21 + OFF8P),
              reduce_u96(d11 + OFF8P), reduce_u96(d22 + OFF8P)}};
    return r;
}
static fp6 fp6_exp_vartime(fp6 x, const u64 *power, int nwords) { /* src/fp6.rs:1072-1084 */
    fp6 res = FP6_ONE;
    for (int w = nwords - 1; w >= 0; w--)
        for (int i = 63; i >= 0; i--) {
            res = fp6_square(res);
            if ((power[w] >> i) & 1) res = fp6_mul(res, x);
        }
    return res;
}
/* Fp6::sqrt, Algorithm 10 of eprint 2012/685 as in src/fp6.rs:922-964 */
static int fp6_sqrt(fp6 self, fp6 *out) {
    static const fp3 E = {{17474221990273267221ULL, 11073203528037158631ULL, 1587630551273719300ULL}};
    static const fp3 F = {{16, 18446744069414584320ULL, 0}};
    static const u64 EXP[1] = {0x3fffffffc0000000ULL};
    fp6 b = fp6_mul(self, fp6_frobenius(self));
    b = fp6_mul(b, fp6_frobenius_double(self));
    b = fp6_exp_vartime(b, EXP, 1);
    fp6 b2 = fp6_square(b);
    fp6 b_pow_p3 = fp6_frobenius_triple(b);
    fp6 check = fp6_mul(b_pow_p3, b);
    fp6 s = fp6_mul(b2, self);
    fp6 sf = fp6_mul_by_low_fp3(s, F);
    fp3 s3 = {{s.c[0], s.c[2], s.c[4]}}, sf3 = {{sf.c[0], sf.c[2], sf.c[4]}};
    fp3 sqrt_s, sqrt_sf;
    if (!fp3_sqrt(s3, &sqrt_s)) sqrt_s = FP3_ZERO;
    if (!fp3_sqrt(sf3, &sqrt_sf)) sqrt_sf = FP3_ZERO;
    int is_one = fp6_eq(check, FP6_ONE);
    fp3 x0 = is_one ? sqrt_s : sqrt_sf;
    fp6 x = fp6_mul_by_low_fp3(b_pow_p3, x0);
    if (!is_one) x = fp6_mul_by_high_fp3(x, E);
    *out = x;
    return fp6_eq(fp6_square(x), self);
}
/* Fp6::exp, src/fp6.rs:1053-1066 (square, multiply, conditional_assign per bit; exp_vartime returns the same element) */
void ora_fp6_exp(const u64 *a, const u64 *power6, u64 *o) {
    fp6 x = fp6_load(a), res = FP6_ONE;
    for (int w = 5; w >= 0; w--)
        for (int i = 63; i >= 0; i--) {
            res = fp6_square(res);
            fp6 tmp = fp6_mul(res, x);
            if ((power6[w] >> i) & 1) res = tmp;
        }
    fp6_store(o, fp6_canonical(res));
}
void ora_fp6_exp_batch(const u64 *a, const u64 *power6, u64 *out, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) ora_fp6_exp(a + 6 * i, power6, out + 6 * i);
}
int ora_fp6_sqrt(const u64 *a, u64 *o) {
    fp6 r;
    int ok = fp6_sqrt(fp6_load(a), &r);
    fp6_store(o, ok ? fp6_canonical(r) : FP6_ZERO);
    return ok;
}
void ora_fp6_sqrt_batch(const u64 *a, u64 *out, uint8_t *is_some, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) is_some[i] = (uint8_t)ora_fp6_sqrt(a + 6 * i, out + 6 * i);
}

/* ------------------------------------------------------------ encodings ---- */

/* Fp::lexicographically_largest, src/fp.rs:281-300 */
static int fp_lex_largest(u64 a) { return fp_canonical(a) >= 0x7fffffff80000001ULL; }
static int fp_is_zero(u64 a) { return fp_canonical(a) == 0; }
/* Fp6::lexicographically_largest, src/fp6.rs:398-424 -- the expression exactly as written there
 * (each term only looks at the coefficient just above it, not at all higher ones). */
static int fp6_lex_largest(fp6 a) {
    return fp_lex_largest(a.c[5]) | (fp_is_zero(a.c[5]) & fp_lex_largest(a.c[4])) | (fp_is_zero(a.c[4]) & fp_lex_largest(a.c[3])) |
           (fp_is_zero(a.c[3]) & fp_lex_largest(a.c[2])) | (fp_is_zero(a.c[2]) & fp_lex_largest(a.c[1])) |
           (fp_is_zero(a.c[1]) & fp_lex_largest(a.c[0]));
}
int ora_fp6_lexicographically_largest(const u64 *a) { return fp6_lex_largest(fp6_load(a)); }

static void fp6_to_bytes(fp6 a, uint8_t *out) { /* src/fp6.rs:1139-1150, src/fp.rs:243-249 */
    for (int i = 0; i < 6; i++) {
        u64 v = fp_canonical(a.c[i]);
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(v >> (8 * b));
    }
}
static int fp6_from_bytes(const uint8_t *in, fp6 *out) { /* src/fp6.rs:1154-1189, src/fp.rs:253-268: none unless every limb < p */
    int ok = 1;
    for (int i = 0; i < 6; i++) {
        u64 v = 0;
        for (int b = 0; b < 8; b++) v |= (u64)in[8 * i + b] << (8 * b);
        int some = v < P_MOD;
        ok &= some;
        out->c[i] = some ? v : 0; /* unwrap_or(Fp::zero()) */
    }
    return ok;
}
/* CompressedPoint::from_affine, src/curve/encoding.rs:55-77 */
void ora_affine_to_compressed(const u64 *xy, uint8_t inf, uint8_t *out49) {
    affine p = affine_load(xy, inf);
    memset(out49, 0, 49);
    fp6_to_bytes(p.inf ? FP6_ZERO : p.x, out49);
    if (p.inf) out49[48] |= 1u << 7;
    if (!p.inf && fp6_lex_largest(p.y)) out49[48] |= 1u << 6;
}
/* UncompressedPoint::from_affine, src/curve/encoding.rs:169-183 */
void ora_affine_to_uncompressed(const u64 *xy, uint8_t inf, uint8_t *out97) {
    affine p = affine_load(xy, inf);
    memset(out97, 0, 97);
    fp6_to_bytes(p.inf ? FP6_ZERO : p.x, out97);
    fp6_to_bytes(p.inf ? FP6_ZERO : p.y, out97 + 48);
    if (p.inf) out97[96] |= 1u << 7;
}
static int affine_is_on_curve(affine p) { /* src/curve/affine.rs:364-368 */
    fp6 lhs = fp6_sub(fp6_square(p.y), fp6_mul(p.x, fp6_add(fp6_square(p.x), FP6_ONE)));
    return fp6_eq(lhs, CURVE_B) | (p.inf != 0);
}
/* UncompressedPoint::to_affine (checked = 0, src/curve/encoding.rs:187-226) and
 * AffinePoint::from_uncompressed (checked = 1: also on the curve and torsion free,
 * src/curve/affine.rs:289-292).  Returns is_some; on none the output is (0, 0, inf = 0). */
int ora_uncompressed_to_affine(const uint8_t *in97, int checked, u64 *out_xy, uint8_t *out_inf) {
    int inf_flag = (in97[96] >> 7) & 1;
    fp6 x, y;
    int ok = fp6_from_bytes(in97, &x) & fp6_from_bytes(in97 + 48, &y);
    affine p = {x, y, (uint8_t)inf_flag};
    if (inf_flag) p = affine_identity();
    ok &= (!inf_flag) | (inf_flag & fp6_is_zero(x) & fp6_is_zero(y));
    if (ok && checked) {
        ok &= affine_is_on_curve(p);
        if (ok) { /* is_torsion_free: JacobianPoint::multiply_vartime(q).is_identity(), src/curve/affine.rs:463-467 */
            static const uint8_t QB[32] = {207, 172, 212, 174, 62, 98, 67, 212, 34, 119, 21, 48, 35, 167, 122, 50,
                                           181, 55, 10, 153, 15, 191, 63, 86, 208, 34, 63, 59, 155, 89, 242, 122};
            proj r = jac_multiply_vartime(proj_from_affine(p), QB);
            ok &= fp6_is_zero(r.z);
        }
    }
    if (!ok) {
        affine z = {FP6_ZERO, FP6_ZERO, 0};
        p = z;
    }
    affine_store(out_xy, out_inf, p);
    return ok;
}

/* CompressedPoint::to_affine (checked = 0: AffinePoint::from_compressed_unchecked, src/curve/encoding.rs:82-129)
 * and AffinePoint::from_compressed (checked = 1: also torsion free, src/curve/affine.rs:300-309). */
int ora_compressed_to_affine(const uint8_t *in49, int checked, u64 *out_xy, uint8_t *out_inf) {
    int inf_flag = (in49[48] >> 7) & 1, sort_flag = (in49[48] >> 6) & 1;
    fp6 x;
    int ok = fp6_from_bytes(in49, &x);
    affine p = affine_identity();
    if (ok) {
        if (inf_flag & !sort_flag & fp6_is_zero(x)) {
            p = affine_identity();
        } else {
            fp6 rhs = fp6_add(fp6_add(fp6_mul(fp6_square(x), x), x), CURVE_B), y;
            ok = fp6_sqrt(rhs, &y);
            if (ok) {
                if (fp6_lex_largest(y) ^ sort_flag) y = fp6_neg(y);
                p.x = x;
                p.y = y;
                p.inf = (uint8_t)inf_flag;
                ok = !inf_flag;
            }
        }
    }
    if (ok && checked) {
        static const uint8_t QB[32] = {207, 172, 212, 174, 62, 98, 67, 212, 34, 119, 21, 48, 35, 167, 122, 50,
                                       181, 55, 10, 153, 15, 191, 63, 86, 208, 34, 63, 59, 155, 89, 242, 122};
        proj r = jac_multiply_vartime(proj_from_affine(p), QB);
        ok &= fp6_is_zero(r.z);
    }
    if (!ok) {
        affine z = {FP6_ZERO, FP6_ZERO, 0};
        p = z;
    }
    affine_store(out_xy, out_inf, p);
    return ok;
}
/* AffinePoint::from_x, src/curve/affine.rs:326-343 */
int ora_affine_from_x(const u64 *x6, u64 *out_xy, uint8_t *out_inf) {
    fp6 x = fp6_load(x6);
    affine p = affine_identity();
    int ok = 1;
    if (!fp6_is_zero(x)) {
        fp6 rhs = fp6_add(fp6_add(fp6_mul(fp6_square(x), x), x), CURVE_B), y;
        ok = fp6_sqrt(rhs, &y);
        if (ok) {
            if (!fp6_lex_largest(y)) y = fp6_neg(y);
            p.x = x;
            p.y = y;
            p.inf = 0;
        }
    }
    if (!ok) {
        affine z = {FP6_ZERO, FP6_ZERO, 0};
        p = z;
    }
    affine_store(out_xy, out_inf, p);
    return ok;
}
void ora_affine_from_x_batch(const u64 *x, u64 *xy, uint8_t *inf, uint8_t *is_some, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) is_some[i] = (uint8_t)ora_affine_from_x(x + 6 * i, xy + 12 * i, inf + i);
}
void ora_decode_compressed_batch(const uint8_t *in, int checked, u64 *xy, uint8_t *inf, uint8_t *is_some, size_t n) {
    ora_init();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) is_some[i] = (uint8_t)ora_compressed_to_affine(in + 49 * i, checked, xy + 12 * i, inf + i);
}
void ora_encode_batch(const u64 *xy, const uint8_t *inf, uint8_t *out, size_t n, int compressed) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        if (compressed) ora_affine_to_compressed(xy + 12 * i, inf[i], out + 49 * i);
        else ora_affine_to_uncompressed(xy + 12 * i, inf[i], out + 97 * i);
    }
}
void ora_decode_uncompressed_batch(const uint8_t *in, int checked, u64 *xy, uint8_t *inf, uint8_t *is_some, size_t n) {
    ora_init();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) is_some[i] = (uint8_t)ora_uncompressed_to_affine(in + 97 * i, checked, xy + 12 * i, inf + i);
}

static void batch_normalize_chunked(const u64 *p, size_t n, size_t chunk, u64 *xy, uint8_t *inf, int jac) {
    if (chunk == 0) chunk = n ? n : 1;
    size_t nchunks = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < nchunks; c++) {
        size_t lo = c * chunk, len = (lo + chunk <= n) ? chunk : n - lo;
        batch_normalize_flat(p + 18 * lo, len, xy + 12 * lo, inf ? inf + lo : NULL, jac);
    }
}
void ora_jac_batch_normalize_chunked(const u64 *p, size_t n, size_t chunk, u64 *xy, uint8_t *inf) {
    batch_normalize_chunked(p, n, chunk, xy, inf, 1);
}
void ora_proj_batch_normalize_chunked(const u64 *p, size_t n, size_t chunk, u64 *xy, uint8_t *inf) {
    batch_normalize_chunked(p, n, chunk, xy, inf, 0);
}
