// Cheetah curve E: y^2 = x^3 + x + (u + 395) over Fp6 -- point formulas.
//
// Every formula follows the reference's operation SEQUENCE (cited per function)
// so that the projective / Jacobian representative (X:Y:Z) it produces is the
// same element-for-element as the reference's, not only the affine point
// (parity level L2 of SURVEY.md 8b).  The field operations underneath are this
// build's own (fp.hip.h, fp6.hip.h).
#pragma once
#include "fp6.hip.h"

namespace cheetah {

struct Affine {  // src/curve/affine.rs:37-41
    Fp6 x, y;
    bool inf;
};
struct Proj {  // src/curve/projective.rs:76-80 (also used for Jacobian, src/curve/jacobian.rs:76-80)
    Fp6 x, y, z;
};
struct MJac {  // src/curve/jacobian.rs:943-948
    Fp6 x, y, z, z4;
};

CH_DEV Affine affine_identity() {  // src/curve/affine.rs:224-230
    Affine a;
    a.x = fp6_zero();
    a.y = fp6_one();
    a.inf = true;
    return a;
}
CH_DEV Proj proj_identity() {  // src/curve/projective.rs:204-210
    Proj p;
    p.x = fp6_zero();
    p.y = fp6_one();
    p.z = fp6_zero();
    return p;
}
CH_DEV Proj proj_from_affine(const Affine& a) {  // src/curve/projective.rs:102-110
    Proj p;
    p.x = a.x;
    p.y = a.y;
    p.z = a.inf ? fp6_zero() : fp6_one();
    return p;
}
CH_DEV MJac mjac_from_affine(const Affine& a) {  // src/curve/jacobian.rs:970-979
    MJac p;
    p.x = a.x;
    p.y = a.y;
    p.z = a.inf ? fp6_zero() : fp6_one();
    p.z4 = p.z;
    return p;
}
// -a; the identity keeps y = 1.                     (src/curve/affine.rs:164-175)
CH_DEV Affine affine_cond_neg(const Affine& a, bool negate) {
    Affine r;
    r.x = a.x;
    r.y = fp6_select(a.y, fp6_select(fp6_neg(a.y), fp6_one(), a.inf), negate);
    r.inf = a.inf;
    return r;
}

// src/curve/projective.rs:400-443.  Same field elements as the reference's sequence; its runs of doublings are
// taken as one small-constant product (8c = 2 (4c), 8 t3^2 instead of 2 (2 t3)^2, 8 b^3 instead of 2 (2b)^2 b).
template <bool INL>
CH_DEV Proj proj_double_unchecked_impl(const Proj& s) {
    auto mul = [](const Fp6& a, const Fp6& b) { return INL ? fp6_mul_inl(a, b) : fp6_mul(a, b); };
    auto sqr = [](const Fp6& a) { return INL ? fp6_square_inl(a) : fp6_square(a); };
    Fp6 a = fp6_triple(sqr(s.x));
    a = fp6_add(a, sqr(s.z));                    // 3 x^2 + z^2
    Fp6 b = mul(s.y, s.z);
    Fp6 t3 = mul(s.y, b);
    Fp6 c = mul(t3, s.x);
    t3 = fp6_mul_small(sqr(t3), 8u);             // 8 (y b)^2
    Fp6 c4 = fp6_mul_small(c, 4u);
    Fp6 d = fp6_sub(sqr(a), fp6_double(c4));     // a^2 - 8c
    Proj r;
    r.x = fp6_double(mul(b, d));
    r.y = fp6_sub(mul(a, fp6_sub(c4, d)), t3);
    r.z = fp6_mul_small(mul(sqr(b), b), 8u);
    return r;
}
CH_DEV Proj proj_double_unchecked(const Proj& s) { return proj_double_unchecked_impl<false>(s); }
// n doublings in a row (the run that opens a ladder window) with the field products inlined: one loop body of about
// 38 KB instead of 12 n calls, whose operands and results cost ~ 29 register moves each at the call boundary.
CH_DEV Proj proj_doubles_unchecked(Proj p, int n) {
#pragma unroll 1
    for (int r = 0; r < n; r++) p = proj_double_unchecked_impl<true>(p);
    return p;
}

// src/curve/projective.rs:349-393 (same values as the unchecked form mod p:
// 4c, 8c, 8 t3^2, 8 b^3; then the identity select)
CH_DEV Proj proj_double(const Proj& s) {
    Proj r = proj_double_unchecked(s);
    if (fp6_is_zero(s.z)) r = proj_identity();
    return r;
}

// Complete addition, Algorithm 1 of Renes-Costello-Batina 2015 as in
// src/curve/projective.rs:456-515.
CH_DEV Proj proj_add(const Proj& s, const Proj& r) {
    Fp6 t0 = fp6_mul(s.x, r.x);
    Fp6 t1 = fp6_mul(s.y, r.y);
    Fp6 t2 = fp6_mul(s.z, r.z);
    Fp6 t3 = fp6_add(s.x, s.y);
    Fp6 t4 = fp6_add(r.x, r.y);
    t3 = fp6_mul(t3, t4);
    t4 = fp6_add(t0, t1);
    t3 = fp6_sub(t3, t4);
    t4 = fp6_add(s.x, s.z);
    Fp6 t5 = fp6_add(r.x, r.z);
    t4 = fp6_mul(t4, t5);
    t5 = fp6_add(t0, t2);
    t4 = fp6_sub(t4, t5);
    t5 = fp6_add(s.y, s.z);
    Fp6 x3 = fp6_add(r.y, r.z);
    t5 = fp6_mul(t5, x3);
    x3 = fp6_add(t1, t2);
    t5 = fp6_sub(t5, x3);
    x3 = fp6_mul_by_3b(t2);
    Fp6 z3 = fp6_add(x3, t4);
    x3 = fp6_sub(t1, z3);
    z3 = fp6_add(t1, z3);
    Fp6 y3 = fp6_mul(x3, z3);
    t1 = fp6_double(t0);
    t1 = fp6_add(t1, t0);
    t4 = fp6_mul_by_3b(t4);
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
    Proj o;
    o.x = x3;
    o.y = y3;
    o.z = z3;
    return o;
}

// src/curve/projective.rs:576-610 (incomplete; returns s when r is the identity)
template <bool INL = false>
CH_DEV Proj proj_add_mixed_unchecked(const Proj& s, const Affine& r) {
    auto mul = [](const Fp6& a, const Fp6& b) { return INL ? fp6_mul_inl(a, b) : fp6_mul(a, b); };
    auto sqr = [](const Fp6& a) { return INL ? fp6_square_inl(a) : fp6_square(a); };
    Fp6 t0 = mul(r.y, s.z);
    t0 = fp6_sub(t0, s.y);
    Fp6 t1 = mul(r.x, s.z);
    t1 = fp6_sub(t1, s.x);
    Fp6 t2 = sqr(t0);
    t2 = mul(t2, s.z);
    Fp6 t3 = sqr(t1);
    Fp6 t4 = mul(t1, t3);
    t2 = fp6_sub(t2, t4);
    Fp6 t5 = mul(t3, s.x);
    Fp6 t6 = fp6_double(t5);
    t2 = fp6_sub(t2, t6);
    Fp6 x3 = mul(t1, t2);
    Fp6 y3 = fp6_sub(t5, t2);
    y3 = mul(y3, t0);
    Fp6 t7 = mul(t4, s.y);
    y3 = fp6_sub(y3, t7);
    Fp6 z3 = mul(t4, s.z);
    Proj o;
    o.x = fp6_select(x3, s.x, r.inf);
    o.y = fp6_select(y3, s.y, r.inf);
    o.z = fp6_select(z3, s.z, r.inf);
    return o;
}

// ModifiedJacobianPoint::double_unchecked (src/curve/jacobian.rs:1094-1128).
// The same field elements are reached with fewer small-constant products: with y2 = 2y,
// 4 y^2 = y2^2, 16 y^4 = (y2^2)^2 = 2c, z3 = y2 z, and c = 8 y^4 is half of 2c.  The products are
// ordered so that at most six Fp6 values are live at once (the ladder kernels run at 168 VGPRs).
template <bool WANT_Z4, bool INL = false>
CH_DEV MJac mjac_double_impl(const MJac& s, bool want_z4_rt = true) {
    auto mul = [](const Fp6& a, const Fp6& b) { return INL ? fp6_mul_inl(a, b) : fp6_mul(a, b); };
    auto sqr = [](const Fp6& a) { return INL ? fp6_square_inl(a) : fp6_square(a); };
    MJac o;
    Fp6 y2 = fp6_double(s.y);
    o.z = mul(y2, s.z);                // z3 = 2 y z
    Fp6 y_sq4 = sqr(y2);               // 4 y^2
    Fp6 c2 = sqr(y_sq4);               // 16 y^4 = 2c
    Fp6 b = fp6_triple(sqr(s.x));
    b = fp6_add(b, s.z4);              // 3 x^2 + z^4
    o.z4 = s.z4;
    if (WANT_Z4 && want_z4_rt) o.z4 = mul(c2, s.z4);  // 2c z^4
    Fp6 c = fp6_halve(c2);             // 8 y^4
    Fp6 a = mul(y_sq4, s.x);           // 4 x y^2
    o.x = sqr(b);
    o.x = fp6_sub(o.x, fp6_double(a));
    o.y = fp6_sub(a, o.x);
    o.y = mul(o.y, b);
    o.y = fp6_sub(o.y, c);
    return o;
}
// src/curve/jacobian.rs:1094-1128
CH_DEV MJac mjac_double_unchecked(const MJac& s) { return mjac_double_impl<true>(s); }
// n doublings in a row followed by an addition (the run that opens a ladder window), with the field products inlined:
// one loop body of about 20 KB instead of 8 n - 1 calls, whose operands and results cost ~ 29 register moves each at
// the call boundary.  The last doubling skips its z4 product (`r < n - 1` is uniform, a scalar branch) and leaves z4
// stale: only valid when the next operation is mjac_add_mixed_unchecked_fresh_z4, which rebuilds z^4 from the z it
// returns, or the point is emitted (a JacobianPoint has no z4, src/curve/jacobian.rs:118-126).  x, y, z are unchanged.
CH_DEV MJac mjac_doubles_then_add(MJac p, int n) {
#pragma unroll 1
    for (int r = 0; r < n; r++) p = mjac_double_impl<true, true>(p, r < n - 1);
    return p;
}

// src/curve/jacobian.rs:1146-1186.  FRESH_Z4: the accumulator's z4 is stale (see above) and z^4 is
// rebuilt from the z that is returned, so it is right for the identity operand too (the reference
// returns self there, whose z4 = z^4 by construction); otherwise the reference's own selection.
// WANT_Z4 = false: the result is emitted, z4 is not needed.  Same products as the reference, ordered
// for short live ranges; each output is selected as soon as it is complete.
template <bool FRESH_Z4, bool WANT_Z4, bool INL = false>
CH_DEV MJac mjac_add_mixed_impl(const MJac& s, const Affine& r) {
    auto mul = [](const Fp6& a, const Fp6& b) { return INL ? fp6_mul_inl(a, b) : fp6_mul(a, b); };
    auto sqr = [](const Fp6& a) { return INL ? fp6_square_inl(a) : fp6_square(a); };
    MJac o;
    Fp6 z1_sq = sqr(s.z);
    Fp6 b = mul(r.x, z1_sq);
    Fp6 d = mul(s.z, z1_sq);
    d = mul(d, r.y);
    Fp6 e = fp6_sub(b, s.x);
    Fp6 f = fp6_sub(d, s.y);
    Fp6 e_sq = sqr(e);
    Fp6 z3 = mul(s.z, e);
    o.z = fp6_select(z3, s.z, r.inf);
    if (!FRESH_Z4) {
        Fp6 z34 = sqr(sqr(z3));
        o.z4 = fp6_select(z34, s.z4, r.inf);
    }
    Fp6 e_sq_e = mul(e, e_sq);
    Fp6 x_e_sq = mul(s.x, e_sq);
    Fp6 x3 = sqr(f);
    x3 = fp6_sub(x3, e_sq_e);
    x3 = fp6_sub(x3, fp6_double(x_e_sq));
    Fp6 y3 = fp6_sub(x_e_sq, x3);
    o.x = fp6_select(x3, s.x, r.inf);
    Fp6 t3 = mul(s.y, e_sq_e);
    y3 = mul(y3, f);
    y3 = fp6_sub(y3, t3);
    o.y = fp6_select(y3, s.y, r.inf);
    if (FRESH_Z4) {
        if (WANT_Z4)
            o.z4 = sqr(sqr(o.z));
        else
            o.z4 = s.z4;
    }
    return o;
}
CH_DEV MJac mjac_add_mixed_unchecked(const MJac& s, const Affine& r) { return mjac_add_mixed_impl<false, true>(s, r); }
template <bool WANT_Z4>
CH_DEV MJac mjac_add_mixed_unchecked_fresh_z4(const MJac& s, const Affine& r) {
    return mjac_add_mixed_impl<true, WANT_Z4>(s, r);
}

// ---- XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2): no counterpart in the reference ----
// Used where only the affine result is returned and the schedule is additions only (k_mul_by_generator_wide): a mixed
// addition is 8 M + 2 S here against the 8 M + 3 S of ModifiedJacobianPoint::add_mixed_unchecked, and there is no
// doubling to pay for the fourth coordinate.  Incomplete like that one (s != +-r, s not the identity); r.inf returns s.
struct XYZZ {
    Fp6 x, y, zz, zzz;
};
CH_DEV XYZZ xyzz_from_affine(const Affine& a) {  // a is not the identity
    XYZZ o;
    o.x = a.x;
    o.y = a.y;
    o.zz = fp6_one();
    o.zzz = fp6_one();
    return o;
}
template <bool INL = false>
CH_DEV XYZZ xyzz_add_mixed_unchecked(const XYZZ& s, const Affine& r) {
    auto mul = [](const Fp6& a, const Fp6& b) { return INL ? fp6_mul_inl(a, b) : fp6_mul(a, b); };
    auto sqr = [](const Fp6& a) { return INL ? fp6_square_inl(a) : fp6_square(a); };
    XYZZ o;
    Fp6 p = fp6_sub(mul(r.x, s.zz), s.x);
    Fp6 rr = fp6_sub(mul(r.y, s.zzz), s.y);
    Fp6 pp = sqr(p);
    Fp6 ppp = mul(p, pp);
    o.zz = fp6_select(mul(s.zz, pp), s.zz, r.inf);
    o.zzz = fp6_select(mul(s.zzz, ppp), s.zzz, r.inf);
    Fp6 q = mul(s.x, pp);
    Fp6 x3 = sqr(rr);
    x3 = fp6_sub(x3, ppp);
    x3 = fp6_sub(x3, fp6_double(q));
    Fp6 y3 = mul(rr, fp6_sub(q, x3));
    o.x = fp6_select(x3, s.x, r.inf);
    y3 = fp6_sub(y3, mul(s.y, ppp));
    o.y = fp6_select(y3, s.y, r.inf);
    return o;
}
CH_DEV Affine affine_from_xyzz(const XYZZ& s) {  // ZZ = 0: the identity
    bool some;
    Fp6 inv = fp6_invert(fp6_mul(s.zz, s.zzz), &some);
    Affine a;
    a.x = fp6_mul(s.x, fp6_mul(inv, s.zzz));
    a.y = fp6_mul(s.y, fp6_mul(inv, s.zz));
    a.inf = false;
    if (!some) a = affine_identity();
    return a;
}

// ---- complete group law (batch entry points cheetah_add_batch / cheetah_double_batch) ----

// ProjectivePoint::add_mixed, src/curve/projective.rs:518-569 (Renes-Costello-Batina Algorithm 2 with Z2 = 1;
// returns s when r is the identity).
CH_DEV Proj proj_add_mixed(const Proj& s, const Affine& r) {
    Fp6 t0 = fp6_mul(s.x, r.x);
    Fp6 t1 = fp6_mul(s.y, r.y);
    Fp6 t3 = fp6_add(r.x, r.y);
    Fp6 t4 = fp6_add(s.x, s.y);
    t3 = fp6_mul(t3, t4);
    t4 = fp6_add(t0, t1);
    t3 = fp6_sub(t3, t4);
    t4 = fp6_mul(r.x, s.z);
    t4 = fp6_add(t4, s.x);
    Fp6 t5 = fp6_mul(r.y, s.z);
    t5 = fp6_add(t5, s.y);
    Fp6 x3 = fp6_mul_by_3b(s.z);
    Fp6 z3 = fp6_add(x3, t4);
    x3 = fp6_sub(t1, z3);
    z3 = fp6_add(t1, z3);
    Fp6 y3 = fp6_mul(x3, z3);
    t1 = fp6_double(t0);
    t1 = fp6_add(t1, t0);
    t4 = fp6_mul_by_3b(t4);
    t1 = fp6_add(t1, s.z);
    Fp6 t2 = fp6_sub(t0, s.z);
    t4 = fp6_add(t4, t2);
    t0 = fp6_mul(t1, t4);
    y3 = fp6_add(y3, t0);
    t0 = fp6_mul(t5, t4);
    x3 = fp6_mul(t3, x3);
    x3 = fp6_sub(x3, t0);
    t0 = fp6_mul(t3, t1);
    z3 = fp6_mul(t5, z3);
    z3 = fp6_add(z3, t0);
    Proj o;
    o.x = fp6_select(x3, s.x, r.inf);
    o.y = fp6_select(y3, s.y, r.inf);
    o.z = fp6_select(z3, s.z, r.inf);
    return o;
}

CH_DEV Proj jac_identity() {  // JacobianPoint::identity() = (1, 1, 0), src/curve/jacobian.rs:223-229
    Proj p;
    p.x = fp6_one();
    p.y = fp6_one();
    p.z = fp6_zero();
    return p;
}

// JacobianPoint::double, src/curve/jacobian.rs:368-402: the modified-Jacobian doubling above on (x, y, z, (z^2)^2)
// reaches the same x3, y3, z3 (z_sq2 there is z4 here); then the identity selection.
CH_DEV Proj jac_double(const Proj& s) {
    MJac m;
    m.x = s.x;
    m.y = s.y;
    m.z = s.z;
    m.z4 = fp6_square(fp6_square(s.z));
    m = mjac_double_impl<false>(m);
    Proj o;
    o.x = m.x;
    o.y = m.y;
    o.z = m.z;
    if (fp6_is_zero(s.z)) o = jac_identity();
    return o;
}

// JacobianPoint::add, src/curve/jacobian.rs:452-505.  The reference's two ct_eq calls (rhs == -self, rhs == self,
// src/curve/jacobian.rs:134-154) compare x z'^2 with x' z^2 and y z'^3 with y' z^3: those are a, b, c, d of the
// formula itself, so the comparisons are e == 0, f == 0 and c + d == 0 and cost no product.  Selections in the
// reference's order.
CH_DEV Proj jac_add(const Proj& s, const Proj& r) {
    Fp6 z1_sq = fp6_square(s.z);
    Fp6 z2_sq = fp6_square(r.z);
    Fp6 a = fp6_mul(s.x, z2_sq);
    Fp6 b = fp6_mul(r.x, z1_sq);
    Fp6 c = fp6_mul(r.z, z2_sq);
    c = fp6_mul(c, s.y);
    Fp6 d = fp6_mul(s.z, z1_sq);
    d = fp6_mul(d, r.y);
    Fp6 e = fp6_sub(b, a);
    Fp6 f = fp6_sub(d, c);
    const bool sz = fp6_is_zero(s.z), rz = fp6_is_zero(r.z);
    const bool same_x = fp6_is_zero(e);
    const bool is_same = (sz && rz) || (!sz && !rz && same_x && fp6_is_zero(f));
    const bool is_neg = (sz && rz) || (!sz && !rz && same_x && fp6_is_zero(fp6_add(c, d)));
    Fp6 e_sq = fp6_square(e);
    Fp6 e_sq_e = fp6_mul(e, e_sq);
    Fp6 a_e_sq = fp6_mul(a, e_sq);
    Proj o;
    o.x = fp6_square(f);
    o.x = fp6_sub(o.x, e_sq_e);
    o.x = fp6_sub(o.x, fp6_double(a_e_sq));
    o.y = fp6_sub(a_e_sq, o.x);
    o.y = fp6_mul(o.y, f);
    o.y = fp6_sub(o.y, fp6_mul(c, e_sq_e));
    o.z = fp6_mul(fp6_mul(s.z, r.z), e);
    if (sz) o = r;
    if (rz) o = s;
    if (is_neg) o = jac_identity();
    if (is_same) o = jac_double(s);
    return o;
}

// JacobianPoint::add_mixed, src/curve/jacobian.rs:508-552.  The reference converts self and -self to affine (one
// Fp6 inversion each, src/curve/affine.rs:86-101) for its two comparisons with rhs; x / z^2 == x' is x == x' z^2
// for z != 0, so here they are e == 0, f == 0 and d + y == 0 on values the formula already has.
CH_DEV Proj jac_add_mixed(const Proj& s, const Affine& r) {
    Fp6 z1_sq = fp6_square(s.z);
    Fp6 b = fp6_mul(r.x, z1_sq);
    Fp6 d = fp6_mul(s.z, z1_sq);
    d = fp6_mul(d, r.y);
    Fp6 e = fp6_sub(b, s.x);
    Fp6 f = fp6_sub(d, s.y);
    const bool sz = fp6_is_zero(s.z);
    const bool same_x = fp6_is_zero(e);
    const bool is_same = (sz && r.inf) || (!sz && !r.inf && same_x && fp6_is_zero(f));
    const bool is_neg = (sz && r.inf) || (!sz && !r.inf && same_x && fp6_is_zero(fp6_add(d, s.y)));
    Fp6 e_sq = fp6_square(e);
    Fp6 e_sq_e = fp6_mul(e, e_sq);
    Fp6 x_e_sq = fp6_mul(s.x, e_sq);
    Proj o;
    o.x = fp6_square(f);
    o.x = fp6_sub(o.x, e_sq_e);
    o.x = fp6_sub(o.x, fp6_double(x_e_sq));
    o.y = fp6_sub(x_e_sq, o.x);
    o.y = fp6_mul(o.y, f);
    o.y = fp6_sub(o.y, fp6_mul(s.y, e_sq_e));
    o.z = fp6_mul(s.z, e);
    if (r.inf) o = s;
    if (is_neg) o = jac_identity();
    if (is_same) o = jac_double(s);
    return o;
}

// JacobianPoint::double_unchecked (src/curve/jacobian.rs:409-439) and ::add_mixed_unchecked (:559-595): the formulas
// above without their identity / equality selections (add_mixed_unchecked keeps "rhs is the identity -> self").
CH_DEV Proj jac_double_unchecked(const Proj& s) {
    MJac m;
    m.x = s.x;
    m.y = s.y;
    m.z = s.z;
    m.z4 = fp6_square(fp6_square(s.z));
    m = mjac_double_impl<false>(m);
    Proj o;
    o.x = m.x;
    o.y = m.y;
    o.z = m.z;
    return o;
}
CH_DEV Proj jac_add_mixed_unchecked(const Proj& s, const Affine& r) {
    MJac m;
    m.x = s.x;
    m.y = s.y;
    m.z = s.z;
    m.z4 = s.z;  // not read: the fresh-z4 form without its z4 output
    m = mjac_add_mixed_impl<true, false>(m, r);
    Proj o;
    o.x = m.x;
    o.y = m.y;
    o.z = m.z;
    return o;
}

// ConstantTimeEq for JacobianPoint (src/curve/jacobian.rs:134-154) and ProjectivePoint (src/curve/projective.rs:118-135)
CH_DEV bool jac_eq(const Proj& s, const Proj& o) {
    const Fp6 z1_sq = fp6_square(s.z), z2_sq = fp6_square(o.z);
    const Fp6 x1 = fp6_mul(s.x, z2_sq), x2 = fp6_mul(o.x, z1_sq);
    const Fp6 y1 = fp6_mul(fp6_mul(s.y, z2_sq), o.z), y2 = fp6_mul(fp6_mul(o.y, z1_sq), s.z);
    const bool sz = fp6_is_zero(s.z), oz = fp6_is_zero(o.z);
    return (sz && oz) || (!sz && !oz && fp6_eq(x1, x2) && fp6_eq(y1, y2));
}
CH_DEV bool proj_eq(const Proj& s, const Proj& o) {
    const Fp6 x1 = fp6_mul(s.x, o.z), x2 = fp6_mul(o.x, s.z);
    const Fp6 y1 = fp6_mul(s.y, o.z), y2 = fp6_mul(o.y, s.z);
    const bool sz = fp6_is_zero(s.z), oz = fp6_is_zero(o.z);
    return (sz && oz) || (!sz && !oz && fp6_eq(x1, x2) && fp6_eq(y1, y2));
}

// Single-point normalisations (one Fp6 inversion each).
// src/curve/affine.rs:64-78
CH_DEV Affine affine_from_proj(const Proj& p) {
    bool some;
    Fp6 zinv = fp6_invert(p.z, &some);
    Affine a;
    a.x = fp6_mul(p.x, zinv);
    a.y = fp6_mul(p.y, zinv);
    a.inf = false;
    if (!some) a = affine_identity();
    return a;
}
// src/curve/affine.rs:86-101
CH_DEV Affine affine_from_jac(const Proj& p) {
    Fp6 z3 = fp6_mul(fp6_square(p.z), p.z);
    bool some;
    Fp6 zinv3 = fp6_invert(z3, &some);
    Affine a;
    a.x = fp6_mul(fp6_mul(p.x, zinv3), p.z);
    a.y = fp6_mul(p.y, zinv3);
    a.inf = false;
    if (!some) a = affine_identity();
    return a;
}

}  // namespace cheetah
