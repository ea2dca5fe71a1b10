// HIP kernels of the batched Cheetah engine (gfx950).  One curve point (or one
// Fp6 element) per lane; 256-thread workgroups = 4 waves = one wave per SIMD.
#pragma once
#include "curve.hip.h"
#include <utility>

namespace cheetah {

typedef uint8_t u8;
typedef int8_t i8;

constexpr int BLOCK = 256;
// Waves per SIMD (= 256-thread workgroups per CU) the kernels are compiled and launched for.
// Measured at 2^20 / 2^24 (tools/ladder_bench.py, tools/bn_bench.py; waves 2 / 3 / 4): constant-time ladder
// 25.6 / 27.8 / 28.4 M/s (at 128 VGPRs the window loop keeps two spill accesses, the rest are in the table
// build), fixed base 91 / 97 / 100 M/s, batch_normalize 7.2 / 7.4 / 7.2 G/s.
#ifndef CH_CT_WAVES
#define CH_CT_WAVES 4
#endif
#ifndef CH_VT_WAVES
#define CH_VT_WAVES 4   // NAF ladder (packed digits: 16 KiB of LDS): 15.8 / 17.3 / 17.8 M/s at 2 / 3 / 4 waves
#endif
#ifndef CH_GEN_WAVES
#define CH_GEN_WAVES 4
#endif
#ifndef CH_VT_PARK
#define CH_VT_PARK 48  // NAF kernels: the wave adds once this many of its 64 lanes are parked at a non-zero digit
#endif
#ifndef CH_MANY_WAVES
// Multi-scalar kernels.  Measured in round 3 (tools/lab/many_lab.py, 2^19 groups, m = 2 constant time / indexed): 2 waves
// per SIMD (226 VGPRs, no spill) 20.4 / 21.6 M groups/s; 3 waves (168 VGPRs, 68 spilled) 23.4 / 24.5; 4 waves (128 VGPRs,
// 124 spilled) 22.4 / 23.6; m = 4: 12.8 / 14.3 / 13.8 (ct).  Keeping the radix-16 digits in LDS, two per byte, instead of
// one byte per digit in global scratch changes nothing at any occupancy (+- 0.2 %), and the window's additions through
// calls instead of inlined are 1 - 3 % slower: what bounds these kernels is the number of resident waves.
#define CH_MANY_WAVES 3
#endif
// NafLookupTable<8> by shared-inversion affine steps (1) or always by the reference's construction (0)
#ifndef CH_NAF_TABLE_AFFINE
#define CH_NAF_TABLE_AFFINE 1
#endif
#ifndef CH_BN_WAVES
#define CH_BN_WAVES 3
#endif

// ---------------------------------------------------------------------------
// Precomputed tables, built on the device at context creation.
struct DeviceTables {
    u64 basepoint[32 * 8 * 12];  // BASEPOINT_TABLE: table-major, entry-major, x|y   src/lookup.rs:136
    u64 shift[12];               // S                                                src/constants.rs:43-61
    u64 minus_shift_4[12];       // MINUS_SHIFT_POINT_ARRAY[4]   = -2^4 S            src/constants.rs:76-90
    u64 minus_shift_256[12];     // MINUS_SHIFT_POINT_ARRAY[256] = -2^256 S
    u64 basepoint_lookup[128 * 12];  // BASEPOINT_LOOKUP: LookupTable<128> = [G .. 128 G], x|y  src/constants.rs:26-27
    u64 odd_multiples[64 * 12];      // ODD_MULTIPLES_BASEPOINT: NafLookupTable<64> = [G, 3G .. 127 G]  src/constants.rs:32-33
    // The accumulator of the radix-16 ladders after their first window's four doublings: every such ladder
    // starts from S and doubles four times before its first addition (src/curve/projective.rs:617-621,
    // src/curve/jacobian.rs:603-607), so these are constants -- the reference's own representatives,
    // computed once with the same formulas.
    u64 shift16_proj[18];            // double_unchecked^4(ProjectivePoint::from(S)): x | y | z
    u64 shift16_mjac[24];            // double_unchecked^4(ModifiedJacobianPoint::from(S)): x | y | z | z4
    // Fixed-base table of k_mul_by_generator_wide (affine results only; no counterpart in the reference):
    // wide[j][e] = (e + 1) 64^j G, window-major, entry-major, x|y.
    u64 wide[43 * 32 * 12];
    // double_unchecked^d(S) for d = 0 .. 256 in both accumulator types, the reference's own representatives: a ladder
    // whose scalar is the same for the whole batch (clear_cofactor, is_torsion_free) starts at the first non-zero digit
    // of its NAF with the accumulator the reference reaches there by doubling S alone.
    u64 shift_dbl_proj[257 * 18];
    u64 shift_dbl_mjac[257 * 24];
    // bit 0: cheetah_ctx_set_point_independent_tables -- the constant-time ladders build EVERY wave's window tables with
    // the reference's complete additions (no wave-level choice that depends on the points)
    u64 flags;
};
constexpr int WIDE_BITS = 6, WIDE_WINDOWS = 43, WIDE_ENTRIES = 32, WIDE_BLOCK = 1024;

// The counter-based generator of cheetah_amd/synth.py (synthetic inputs, seeded random points).
CH_DEV u64 splitmix64(u64 seed, u64 idx) {
    u64 z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---- AoS <-> register helpers --------------------------------------------
CH_DEV Fp6 load_fp6(const u64* p) {
    Fp6 r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.c[i] = p[i];
    return r;
}
CH_DEV void store_fp6_canonical(u64* p, const Fp6& a) {
#pragma unroll
    for (int i = 0; i < 6; i++) p[i] = fp_canonical(a.c[i]);
}
CH_DEV Affine load_affine(const u64* p) {
    Affine a;
    a.x = load_fp6(p);
    a.y = load_fp6(p + 6);
    a.inf = false;
    return a;
}
CH_DEV Proj load_proj(const u64* p) {
    Proj r;
    r.x = load_fp6(p);
    r.y = load_fp6(p + 6);
    r.z = load_fp6(p + 12);
    return r;
}
CH_DEV void store_affine(u64* xy, u8* inf, size_t i, const Affine& a) {
    store_fp6_canonical(xy + 12 * i, a.x);
    store_fp6_canonical(xy + 12 * i + 6, a.y);
    if (inf) inf[i] = a.inf ? 1 : 0;
}
CH_DEV void store_proj(u64* out, u8* inf, size_t i, const Proj& p) {
    store_fp6_canonical(out + 18 * i, p.x);
    store_fp6_canonical(out + 18 * i + 6, p.y);
    store_fp6_canonical(out + 18 * i + 12, p.z);
    if (inf) inf[i] = fp6_is_zero(p.z) ? 1 : 0;
}
// out_form 0: normalise (one Fp6 inversion per lane), 1: the projective representative as is
// out_form 3: only the identity flag (is_identity() of the representative, z == 0); 4: AND it into inf[i]
template <int OUT_FORM>
CH_DEV void emit_point(u64* out, u8* inf, size_t i, const Proj& p) {
    if (OUT_FORM == 3)
        inf[i] = fp6_is_zero(p.z) ? 1 : 0;
    else if (OUT_FORM == 4)
        inf[i] = (inf[i] && fp6_is_zero(p.z)) ? 1 : 0;
    else if (OUT_FORM == 0)
        store_affine(out, inf, i, affine_from_proj(p));
    else
        store_proj(out, inf, i, p);
}
// Modified-Jacobian accumulator: out_form 0 = AffinePoint::from(&JacobianPoint)
// (src/curve/affine.rs:86-101), 2 = the Jacobian representative (x, y, z) as is
// (From<ModifiedJacobianPoint> for JacobianPoint drops z^4, src/curve/jacobian.rs:118-126).
template <int OUT_FORM>
CH_DEV void emit_point(u64* out, u8* inf, size_t i, const MJac& p) {
    Proj j;
    j.x = p.x;
    j.y = p.y;
    j.z = p.z;
    if (OUT_FORM == 3)
        inf[i] = fp6_is_zero(p.z) ? 1 : 0;
    else if (OUT_FORM == 4)  // flags[i] &= is_identity: the subgroup check of AffinePoint::from_uncompressed
        inf[i] = (inf[i] && fp6_is_zero(p.z)) ? 1 : 0;
    else if (OUT_FORM == 0)
        store_affine(out, inf, i, affine_from_jac(j));
    else
        store_proj(out, inf, i, j);
}

// Ladder accumulators: the two coordinate systems the reference runs its ladders
// in.  Proj = ProjectivePoint (src/curve/projective.rs:400-443, :576-610), MJac =
// ModifiedJacobianPoint (src/curve/jacobian.rs:1094-1128, :1146-1186).
template <class ACC> struct Ladder;
// dbls_then_add / add_after_dbl / add_last: the run of doublings that opens a window (field products inlined,
// curve.hip.h), the addition that follows it, and the ladder's closing addition.  For MJac the last doubling of
// the run skips its z4 product, which the addition would overwrite anyway (it rebuilds z^4 from z), and the
// closing addition skips z4 altogether: x, y, z -- everything a JacobianPoint has -- are untouched.
template <> struct Ladder<Proj> {
    static CH_DEV Proj from_affine(const Affine& a) { return proj_from_affine(a); }
    static CH_DEV Proj shift16(const DeviceTables* tab) { return load_proj(tab->shift16_proj); }
    static CH_DEV Proj shift_doubled(const DeviceTables* tab, int d) { return load_proj(tab->shift_dbl_proj + 18 * d); }
    static CH_DEV Proj dbl_inl(const Proj& p) { return proj_double_unchecked_impl<true>(p); }  // field products inlined
    static CH_DEV Proj add(const Proj& p, const Affine& q) { return proj_add_mixed_unchecked(p, q); }
    static CH_DEV Proj dbls_then_add(const Proj& p, int n) { return proj_doubles_unchecked(p, n); }
    static CH_DEV Proj add_after_dbl(const Proj& p, const Affine& q) { return proj_add_mixed_unchecked(p, q); }
    static CH_DEV Proj add_last(const Proj& p, const Affine& q) { return proj_add_mixed_unchecked(p, q); }
    static CH_DEV Proj add_inl(const Proj& p, const Affine& q) { return proj_add_mixed_unchecked<true>(p, q); }
};
template <> struct Ladder<MJac> {
    static CH_DEV MJac from_affine(const Affine& a) { return mjac_from_affine(a); }
    static CH_DEV MJac shift16(const DeviceTables* tab) {
        MJac m;
        m.x = load_fp6(tab->shift16_mjac);
        m.y = load_fp6(tab->shift16_mjac + 6);
        m.z = load_fp6(tab->shift16_mjac + 12);
        m.z4 = load_fp6(tab->shift16_mjac + 18);
        return m;
    }
    static CH_DEV MJac shift_doubled(const DeviceTables* tab, int d) {
        const u64* o = tab->shift_dbl_mjac + 24 * d;
        MJac m;
        m.x = load_fp6(o);
        m.y = load_fp6(o + 6);
        m.z = load_fp6(o + 12);
        m.z4 = load_fp6(o + 18);
        return m;
    }
    static CH_DEV MJac dbl_inl(const MJac& p) { return mjac_double_impl<true, true>(p); }
    static CH_DEV MJac dbls_then_add(const MJac& p, int n) { return mjac_doubles_then_add(p, n); }
    static CH_DEV MJac add(const MJac& p, const Affine& q) { return mjac_add_mixed_unchecked(p, q); }
    static CH_DEV MJac add_after_dbl(const MJac& p, const Affine& q) { return mjac_add_mixed_unchecked_fresh_z4<true>(p, q); }
    static CH_DEV MJac add_last(const MJac& p, const Affine& q) { return mjac_add_mixed_unchecked_fresh_z4<false>(p, q); }
    // any addition, z^4 rebuilt from the z returned, field products inlined
    static CH_DEV MJac add_inl(const MJac& p, const Affine& q) { return mjac_add_mixed_impl<true, true, true>(p, q); }
};

CH_DEV Fp6 curve_b() {  // B = u + 395, src/curve/mod.rs:42-49
    Fp6 b = fp6_zero();
    b.c[0] = 395;
    b.c[1] = 1;
    return b;
}

// (0, 1) is the affine encoding of the identity (src/curve/affine.rs:224-230); it is not a point of the curve
// (1 != B), so the encoding is unambiguous.
CH_DEV bool affine_is_identity_encoding(const Fp6& x, const Fp6& y) {  // branch-free, like fp6_is_zero
    u64 acc = fp_canonical(y.c[0]) ^ 1ull;
#pragma unroll
    for (int i = 0; i < 6; i++) acc |= fp_nonzero_bits(x.c[i]);
#pragma unroll
    for (int i = 1; i < 6; i++) acc |= fp_nonzero_bits(y.c[i]);
    return acc == 0;
}

// is_on_curve of the three point types as device predicates (src/curve/affine.rs:364-368, projective.rs:940-946,
// jacobian.rs:930-938); the identity (its (0, 1) encoding, z == 0) counts as on the curve, as there.
CH_DEV bool affine_on_curve(const Fp6& x, const Fp6& y) {
    const Fp6 lhs = fp6_sub(fp6_square(y), fp6_mul(x, fp6_add(fp6_square(x), fp6_one())));  // y^2 - x (x^2 + 1) == B
    return fp6_eq(lhs, curve_b());
}
template <int FORM>
CH_DEV bool point_on_curve(const u64* pts, size_t i) {
    if (FORM == 0) {
        const Affine p = load_affine(pts + 12 * i);
        return affine_on_curve(p.x, p.y) || affine_is_identity_encoding(p.x, p.y);
    }
    const Proj p = load_proj(pts + 18 * i);
    const Fp6 x3 = fp6_mul(fp6_square(p.x), p.x), z2 = fp6_square(p.z);
    if (FORM == 1) {
        const Fp6 rhs = fp6_add(fp6_add(x3, fp6_mul(p.x, z2)), fp6_mul(fp6_mul(z2, p.z), curve_b()));
        return fp6_eq(fp6_mul(fp6_square(p.y), p.z), rhs) || fp6_is_zero(p.z);
    }
    const Fp6 z4 = fp6_square(z2);
    const Fp6 rhs = fp6_add(fp6_add(x3, fp6_mul(p.x, z4)), fp6_mul(fp6_mul(z2, z4), curve_b()));
    return fp6_eq(fp6_square(p.y), rhs) || fp6_is_zero(p.z);
}

// The point whose multiples fill the window table, as a projective point:
//   affine input     ProjectivePoint::from(&AffinePoint)              src/curve/projective.rs:102-110
//   projective input as is
//   Jacobian input   LookupTable::from(&JacobianPoint) normalises first: AffinePoint::from(p),
//                    one Fp6 inversion                                src/lookup.rs:45-49, src/curve/affine.rs:86-101
template <int IN_FORM>
CH_DEV Proj load_input_point(const u64* pts, size_t i) {
    if (IN_FORM == 0) {
        Affine a = load_affine(pts + 12 * i);
        a.inf = affine_is_identity_encoding(a.x, a.y);
        return proj_from_affine(a);
    }
    if (IN_FORM == 2) return proj_from_affine(affine_from_jac(load_proj(pts + 18 * i)));
    return load_proj(pts + 18 * i);
}

// ---------------------------------------------------------------------------
// Table construction (context creation).  Thread 32 builds -2^4 S and -2^256 S
// (src/constants.rs:80-90); thread 64 + e builds entry (e + 1) G of
// BASEPOINT_LOOKUP (src/constants.rs:26-27); thread 192 + e builds entry (2e + 1) G of
// ODD_MULTIPLES_BASEPOINT (src/constants.rs:32-33).  Entries are affine, hence
// independent of how they were computed.
constexpr int BUILD_TABLES_THREADS = 256;
__global__ void __launch_bounds__(BUILD_TABLES_THREADS) k_build_tables(DeviceTables* tab, const u64* gen_xy, const u64* shift_xy) {
    const int t = threadIdx.x;
    if (t == 0) tab->flags = 0;
    if (t < 32) {
        // BASEPOINT_TABLE itself is built by k_build_comb_table (shared with BasePointTable::create for any base)
    } else if (t == 32) {
        Affine s = load_affine(shift_xy);
        store_affine(tab->shift, nullptr, 0, s);
        Proj cur = proj_from_affine(s);
        cur.y = fp6_neg(cur.y);
        for (int i = 1; i <= 256; i++) {
            cur = proj_double(cur);
            if (i == 4) store_affine(tab->minus_shift_4, nullptr, 0, affine_from_proj(cur));
        }
        store_affine(tab->minus_shift_256, nullptr, 0, affine_from_proj(cur));
    } else if (t == 33) {
        const Affine s = load_affine(shift_xy);
        Proj p = proj_from_affine(s);
        MJac m = mjac_from_affine(s);
        for (int i = 0; i < 4; i++) {
            p = proj_double_unchecked(p);
            m = mjac_double_unchecked(m);
        }
        store_proj(tab->shift16_proj, nullptr, 0, p);
        store_fp6_canonical(tab->shift16_mjac, m.x);
        store_fp6_canonical(tab->shift16_mjac + 6, m.y);
        store_fp6_canonical(tab->shift16_mjac + 12, m.z);
        store_fp6_canonical(tab->shift16_mjac + 18, m.z4);
    } else if (t == 34 || t == 35) {
        const Affine s = load_affine(shift_xy);
        Proj p = proj_from_affine(s);
        MJac m = mjac_from_affine(s);
        for (int d = 0; d <= 256; d++) {
            if (t == 34) {
                store_proj(tab->shift_dbl_proj, nullptr, (size_t)d, p);
                p = proj_double_unchecked(p);
            } else {
                u64* o = tab->shift_dbl_mjac + 24 * d;
                store_fp6_canonical(o, m.x);
                store_fp6_canonical(o + 6, m.y);
                store_fp6_canonical(o + 12, m.z);
                store_fp6_canonical(o + 18, m.z4);
                m = mjac_double_unchecked(m);
            }
        }
    } else if (t >= 64) {
        const bool odd = t >= 192;
        const int m = odd ? 2 * (t - 192) + 1 : t - 63;  // multiplier, by left-to-right double-and-add with the complete formulas
        const Proj g = proj_from_affine(load_affine(gen_xy));
        Proj acc = proj_identity();
        for (int b = 7; b >= 0; b--) {
            acc = proj_double(acc);
            Proj sum = proj_add(acc, g);
            if ((m >> b) & 1) acc = sum;
        }
        if (odd)
            store_affine(tab->odd_multiples, nullptr, (size_t)(t - 192), affine_from_proj(acc));
        else
            store_affine(tab->basepoint_lookup, nullptr, (size_t)(m - 1), affine_from_proj(acc));
    }
}

// BasePointTable::create (src/lookup.rs:164-173): thread t builds LookupTable<8> of 256^t P (:25-37): eight
// multiples by the complete addition, affine (identities stored as (0, 1)).  Entries are affine, hence independent
// of how they were computed.  Used for BASEPOINT_TABLE (P = G) and for caller-supplied bases.
template <int IN_FORM>
__global__ void __launch_bounds__(64) k_build_comb_table(u64* comb /* [32][8][12] */, const u64* base_pt) {
    const int t = threadIdx.x;
    if (t >= 32) return;
    Proj base = load_input_point<IN_FORM>(base_pt, 0);
    for (int i = 0; i < 8 * t; i++) base = proj_double(base);
    Proj cur = base;
    for (int j = 0; j < 8; j++) {
        if (j > 0) cur = proj_add(base, cur);
        store_affine(comb, nullptr, (size_t)(8 * t + j), affine_from_proj(cur));
    }
}

// flag[0] = 1 iff no entry of a comb table is the identity or has y = 0.  Such an entry means that the base has small
// order or that the reference's chain of complete additions met its exceptional pair (quirk Q7): the reference's table
// is then NOT the table of true multiples, and the widened comb, built from true multiples, must not stand in for it.
__global__ void __launch_bounds__(256) k_comb_table_is_ordinary(const u64* __restrict__ comb /* [256][12] */, u8* __restrict__ flag) {
    const Affine a = load_affine(comb + 12 * threadIdx.x);
    const int odd = (int)affine_is_identity_encoding(a.x, a.y) | (int)fp6_is_zero(a.y);
    const int any = __syncthreads_or(odd);
    if (threadIdx.x == 0) flag[0] = any ? 0 : 1;
}

// ---------------------------------------------------------------------------
// Elementwise Fp6 kernels (config 1): HBM-bound, 144 B per multiplication.
// An element is 48 contiguous bytes = three 16-byte words; lane i of a wave reads
// words 3i, 3i+1, 3i+2 of the wave's 3 KiB span with three global_load_dwordx4.
// Every 128-byte line of the span is touched by all three instructions, which
// issue back to back, so the line is fetched from HBM once and served from the
// CU's L1 after that; no LDS staging and no barriers, which leaves the waves
// free to overlap one element's loads with another's arithmetic.
enum { EW_MUL = 0, EW_SQUARE = 1, EW_INVERT = 2, EW_SQRT = 3 };

typedef u64 u64x2 __attribute__((ext_vector_type(2)));

CH_DEV Fp6 load_fp6_vec(const u64* p) {  // p is 16-byte aligned
#ifdef CH_BN_NOMEM  // experiment: no memory reads at all, values derived from the address
    Fp6 f;
    for (int i = 0; i < 6; i++) f.c[i] = (u64)(size_t)p * 0x9E3779B97F4A7C15ull + i;
    return f;
#endif
    const u64x2* q = reinterpret_cast<const u64x2*>(p);
    u64x2 v0 = q[0], v1 = q[1], v2 = q[2];
    Fp6 r;
    r.c[0] = v0.x; r.c[1] = v0.y; r.c[2] = v1.x; r.c[3] = v1.y; r.c[4] = v2.x; r.c[5] = v2.y;
    return r;
}
CH_DEV void store_fp6_vec_canonical(u64* p, const Fp6& a) {
#ifdef CH_BN_NOMEM
    if (a.c[0] == 0x1234567 && a.c[3] == 0x7654321) p[0] = a.c[1];  // practically never: keeps the value alive
    return;
#endif
    u64x2* q = reinterpret_cast<u64x2*>(p);
    u64x2 v0, v1, v2;
    v0.x = fp_canonical(a.c[0]); v0.y = fp_canonical(a.c[1]);
    v1.x = fp_canonical(a.c[2]); v1.y = fp_canonical(a.c[3]);
    v2.x = fp_canonical(a.c[4]); v2.y = fp_canonical(a.c[5]);
    q[0] = v0; q[1] = v1; q[2] = v2;
}

CH_DEV void store_fp6_vec_raw(u64* p, const Fp6& a) {  // limbs as they are (internal state, not an output)
    u64x2* q = reinterpret_cast<u64x2*>(p);
    u64x2 v0, v1, v2;
    v0.x = a.c[0]; v0.y = a.c[1]; v1.x = a.c[2]; v1.y = a.c[3]; v2.x = a.c[4]; v2.y = a.c[5];
    q[0] = v0; q[1] = v1; q[2] = v2;
}

template <int OP>
__global__ void __launch_bounds__(BLOCK) k_fp6_elementwise(const u64* __restrict__ a, const u64* __restrict__ b,
                                                           u64* __restrict__ out, u8* __restrict__ is_some, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const Fp6 x = load_fp6_vec(a + 6 * i);
        Fp6 r;
        bool some = true;
        if (OP == EW_MUL)
            r = fp6_mul(x, load_fp6_vec(b + 6 * i));
        else if (OP == EW_SQUARE)
            r = fp6_square(x);
        else if (OP == EW_INVERT)
            r = fp6_invert(x, &some);
        else {  // CtOption::none leaves as 0 with is_some = 0
            r = fp6_sqrt(x, &some);
            r = fp6_select(fp6_zero(), r, some);
        }
        store_fp6_vec_canonical(out + 6 * i, r);
        if ((OP == EW_INVERT || OP == EW_SQRT) && is_some) is_some[i] = some ? 1 : 0;
    }
}

// The coefficient-wise and structured Fp6 operations over a batch (one public entry point, cheetah_fp6_op_batch):
// add / sub / neg / double / triple (src/fp6.rs:968-1029), mul_by_u32 / mul_by_fp (:428-464), make_canonical (:373-382),
// the Frobenius maps (:1088-1123) and mul_by_3b (src/curve/mod.rs:54-106).  HBM-bound: the op is a run-time switch.
enum Fp6Op { F6_ADD = 0, F6_SUB, F6_NEG, F6_DOUBLE, F6_TRIPLE, F6_MUL_BY_U32, F6_MUL_BY_FP, F6_CANONICAL, F6_FROBENIUS,
             F6_FROBENIUS_DOUBLE, F6_FROBENIUS_TRIPLE, F6_MUL_BY_3B, F6_OPS };
__global__ void __launch_bounds__(BLOCK) k_fp6_op(int op, const u64* __restrict__ a, const u64* __restrict__ b, u64 scalar,
                                                  u64* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const Fp6 x = load_fp6_vec(a + 6 * i);
        Fp6 r;
        switch (op) {
            case F6_ADD: r = fp6_add(x, load_fp6_vec(b + 6 * i)); break;
            case F6_SUB: r = fp6_sub(x, load_fp6_vec(b + 6 * i)); break;
            case F6_NEG: r = fp6_neg(x); break;
            case F6_DOUBLE: r = fp6_double(x); break;
            case F6_TRIPLE: r = fp6_triple(x); break;
            case F6_MUL_BY_U32: r = fp6_mul_by_fp(x, scalar & 0xffffffffull); break;  // the same element of Fp as Fp::mul_by_u32 gives
            case F6_MUL_BY_FP: r = fp6_mul_by_fp(x, scalar); break;
            case F6_FROBENIUS: r = fp6_frobenius(x); break;
            case F6_FROBENIUS_DOUBLE: r = fp6_frobenius_double(x); break;
            case F6_FROBENIUS_TRIPLE: r = fp6_frobenius_triple(x); break;
            case F6_MUL_BY_3B: r = fp6_mul_by_3b(x); break;
            default: r = x; break;  // F6_CANONICAL: the store canonicalises
        }
        store_fp6_vec_canonical(out + 6 * i, r);
    }
}

// Fp6::exp / exp_vartime over a batch, one exponent for the whole batch (cheetah_fp6_exp_batch).
__global__ void __launch_bounds__(BLOCK) k_fp6_exp(const u64* __restrict__ a, u64 p0, u64 p1, u64 p2, u64 p3, u64 p4, u64 p5,
                                                   u64* __restrict__ out, size_t n) {
    const u64 power[6] = {p0, p1, p2, p3, p4, p5};
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK)
        store_fp6_vec_canonical(out + 6 * i, fp6_exp(load_fp6_vec(a + 6 * i), power));
}

// The base-field operations over a batch (cheetah_fp_op_batch): Fp::add / sub / neg / double / triple / mul / square /
// mul_by_u32 / invert / make_canonical (src/fp.rs:125-196, :335-359).  One element per lane, HBM-bound except invert.
enum FpOp { FP_ADD = 0, FP_SUB, FP_NEG, FP_DOUBLE, FP_TRIPLE, FP_MUL, FP_SQUARE, FP_MUL_BY_U32, FP_INVERT, FP_CANONICAL, FP_SQRT, FP_EXP,
            FP_OPS };
__global__ void __launch_bounds__(BLOCK) k_fp_op(int op, const u64* __restrict__ a, const u64* __restrict__ b, u64 scalar,
                                                 u64* __restrict__ out, u8* __restrict__ is_some, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const u64 x = a[i];
        u64 r;
        switch (op) {
            case FP_ADD: r = fp_add(x, b[i]); break;
            case FP_SUB: r = fp_sub(x, b[i]); break;
            case FP_NEG: r = fp_neg(x); break;
            case FP_DOUBLE: r = fp_double(x); break;
            case FP_TRIPLE: r = fp_triple(x); break;
            case FP_MUL: r = fp_mul(x, b[i]); break;
            case FP_SQUARE: r = fp_square(x); break;
            case FP_MUL_BY_U32: r = fp_mul(x, scalar & 0xffffffffull); break;
            case FP_INVERT: r = fp_invert(x); break;  // 0 -> 0, flagged none (CtOption, src/fp.rs:335-359)
            case FP_SQRT: {                           // Fp::sqrt, src/fp.rs:199-229; a non-square leaves as 0 with is_some = 0
                bool some;
                r = fp_sqrt(x, &some);
                if (!some) r = 0;
                if (is_some) is_some[i] = some ? 1 : 0;
                break;
            }
            case FP_EXP: r = fp_exp(x, scalar); break;  // Fp::exp / exp_vartime, src/fp.rs:304-333
            default: r = x; break;
        }
        out[i] = fp_canonical(r);
        if (op == FP_INVERT && is_some) is_some[i] = fp_is_zero(x) ? 0 : 1;
    }
}

// ---------------------------------------------------------------------------
// batch_normalize (config 4).  The reference runs ONE sequential Montgomery
// chain over the whole slice (src/curve/projective.rs:897-930,
// src/curve/jacobian.rs:886-920).  Here every lane runs the same trick over its
// own CHUNK points: prefix products of the non-zero z's, one Fp6 inversion per
// lane, then the backward pass.  Identities (z = 0) are skipped in the chain and
// come out as (0, 1, inf = 1) exactly as in the reference.  Exact field
// arithmetic makes the re-association invisible in the results.
// A lane's chunk is strided: lane t of a workgroup owns points
// tile + t, tile + BLOCK + t, ..., so at every step the workgroup touches BLOCK
// consecutive 144-byte records with 16-byte accesses -- each 128-byte line is
// fetched from HBM once -- instead of one record per 1 152 bytes.
// Measured (MI355X, 2^24 Jacobian points, tools/bn_bench.py, profiles/r02_bn_variants.txt).  The first form of this
// kernel was fully unrolled over the chunk with the prefixes in registers and every field product a call: 7.35 G
// points/s, 8.0 once each step's operands were requested ahead of their use; chunk sizes 4 - 12, prefixes in LDS,
// 2 - 4 waves per SIMD, deeper prefetch (global_load_lds_dwordx4 rings) all measured the same 2.06 - 2.12 ms, against
// 1.80 ms for the arithmetic alone (-DCH_BN_NOMEM) and 0.87 ms for the records alone (tools/mem_pattern.hip).
// What moved it was the instruction count: both passes are now ROLLED loops, one body each, with the field products
// inlined (fp6_mul_inl: no calls, no operand marshalling; the unrolled form inlined would be ~150 KB of code).  A
// rolled loop cannot index a register array, so the prefix products before points 2 .. CHUNK - 1 live in LDS
// ([slot][limb][lane], 8-byte accesses, conflict free; 48 KiB per workgroup) and the one before point 1 in registers;
// the next point's operands are still requested one step ahead (z two steps ahead in the forward pass).  Shapes
// measured on one box (default before: 2.158 ms): chunk 6 at 3 waves per SIMD 1.978 ms (kept), chunk 7 with two
// register prefixes at 3 waves 1.982, chunk 8 at 2 waves 1.993 - 2.02, chunk 5 - 6 at 4 waves 2.13.
#ifndef CH_BN_CHUNK
#define CH_BN_CHUNK 6
#endif
#ifndef CH_BN_NREG
#define CH_BN_NREG 1
#endif
#ifndef CH_BN_FASTPATH
#define CH_BN_FASTPATH 1   // wave-uniform skip of the identity selects (round 3; A/B in tools/lab/bn_lab.py)
#endif
template <int JAC>
__global__ void __launch_bounds__(BLOCK, CH_BN_WAVES) k_batch_normalize(const u64* __restrict__ xyz, u64* __restrict__ xy,
                                                                     u8* __restrict__ inf, size_t n) {
    constexpr int CHUNK = CH_BN_CHUNK;
    constexpr int NREG = CH_BN_NREG;  // prefixes before points 1 .. NREG in registers, the others in LDS
    __shared__ u64 s_prefix[(CHUNK - 1 - NREG) * 6 * BLOCK];
    const int t = threadIdx.x;
    const size_t tile_pts = (size_t)BLOCK * CHUNK;
    const size_t ntiles = (n + tile_pts - 1) / tile_pts;
    auto load_z = [&](size_t e) { return e < n ? load_fp6_vec(xyz + 18 * e + 12) : fp6_zero(); };
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t e0 = tile * tile_pts + t;
        Fp6 z = load_z(e0), za = load_z(e0 + BLOCK), zb = load_z(e0 + 2 * (size_t)BLOCK);
        Fp6 acc = fp6_select(z, fp6_one(), fp6_is_zero(z));
        const Fp6 pre1 = acc;
        Fp6 pre2 = acc;
#pragma unroll 1
        for (int j = 1; j < CHUNK; j++) {
            z = za;
            za = zb;
            zb = j + 2 < CHUNK ? load_z(e0 + (size_t)(j + 2) * BLOCK) : fp6_zero();
            if (NREG == 2 && j == 2) pre2 = acc;
            if (j > NREG) {
#pragma unroll
                for (int l = 0; l < 6; l++) s_prefix[((j - 1 - NREG) * 6 + l) * BLOCK + t] = acc.c[l];
            }
            const Fp6 m = fp6_mul_inl(acc, z);
#if CH_BN_FASTPATH
            const bool zz = fp6_is_zero(z);
            if (__any(zz)) acc = fp6_select(m, acc, zz);  // identities are rare: the selects run only in a wave that holds one
            else acc = m;
#else
            acc = fp6_select(m, acc, fp6_is_zero(z));
#endif
        }
        bool some;
        acc = fp6_invert(acc, &some);
        Fp6 nx, ny, nz;
        {
            const size_t e = e0 + (size_t)(CHUNK - 1) * BLOCK;
            const bool in = e < n;
            const u64* p = xyz + 18 * (in ? e : 0);
            nx = load_fp6_vec(p);
            ny = load_fp6_vec(p + 6);
            nz = in ? load_fp6_vec(p + 12) : fp6_zero();
        }
#pragma unroll 1
        for (int j = CHUNK - 1; j >= 0; j--) {
            const size_t e = e0 + (size_t)j * BLOCK;
            const Fp6 x = nx, y = ny;
            z = nz;
            if (j > 0) {
                const size_t ep = e - BLOCK;
                const bool in = ep < n;
                const u64* p = xyz + 18 * (in ? ep : 0);
                nx = load_fp6_vec(p);
                ny = load_fp6_vec(p + 6);
                nz = in ? load_fp6_vec(p + 12) : fp6_zero();
            }
            const bool skip = fp6_is_zero(z);
#if CH_BN_FASTPATH
            const bool any_skip = __any(skip);
#endif
            Fp6 zi = acc;  // 1 / z_j
            if (j > 0) {
                Fp6 pre = (NREG == 2 && j == 2) ? pre2 : pre1;
                if (j > NREG) {
#pragma unroll
                    for (int l = 0; l < 6; l++) pre.c[l] = s_prefix[((j - 1 - NREG) * 6 + l) * BLOCK + t];
                }
                zi = fp6_mul_inl(pre, acc);
                const Fp6 m = fp6_mul_inl(acc, z);
#if CH_BN_FASTPATH
                if (any_skip) acc = fp6_select(m, acc, skip);
                else acc = m;
#else
                acc = fp6_select(m, acc, skip);
#endif
            }
            Affine o;
            if (JAC) {
                const Fp6 zi2 = fp6_square_inl(zi);
                o.x = fp6_mul_inl(x, zi2);
                o.y = fp6_mul_inl(fp6_mul_inl(y, zi2), zi);
            } else {
                o.x = fp6_mul_inl(x, zi);
                o.y = fp6_mul_inl(y, zi);
            }
            o.inf = false;
#if CH_BN_FASTPATH
            if (any_skip) {
                o.x = fp6_select(o.x, fp6_zero(), skip);
                o.y = fp6_select(o.y, fp6_one(), skip);
                o.inf = skip;
            }
#else
            if (skip) o = affine_identity();
#endif
            if (e < n) {
                store_fp6_vec_canonical(xy + 12 * e, o.x);
                store_fp6_vec_canonical(xy + 12 * e + 6, o.y);
                inf[e] = o.inf ? 1 : 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Scalar recoding (src/scalar.rs:531-548): 32 bytes -> 64 signed radix-16 digits
// in [-8, 8) (digit 63 is left un-recentred, up to 16).  Digits go to LDS,
// digit-major, one byte per lane.
CH_DEV void recode_radix16(const u64 k[4], i8* digits /* [64][BLOCK] */, int t) {
    int carry = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        u64 v = k[w];
        for (int j = 0; j < 16; j++) {
            const int i = 16 * w + j;
            int d = (int)(v & 0xf) + carry;
            v >>= 4;
            if (i < 63) {
                carry = (d + 8) >> 4;
                d -= carry << 4;
            }
            digits[i * BLOCK + t] = (i8)d;
        }
    }
}

// Width-w NAF (src/scalar.rs:585-643), one digit per bit position, to LDS.
CH_DEV void recode_wnaf(const u64 k[4], int w, i8* naf /* [256][BLOCK] */, int t) {
    for (int i = 0; i < 256; i++) naf[i * BLOCK + t] = 0;
    const u32 width = 1u << w, mask = width - 1;
    u32 carry = 0;
    int pos = 0;
    while (pos < 256) {
        const int idx = pos >> 6, bit = pos & 63;
        // x_u64[idx] with the reference's fifth, zero word
        u64 lo = idx == 0 ? k[0] : idx == 1 ? k[1] : idx == 2 ? k[2] : k[3];
        u64 hi = idx == 0 ? k[1] : idx == 1 ? k[2] : idx == 2 ? k[3] : 0ull;
        u64 buf = (bit < 64 - w) ? (lo >> bit) : ((lo >> bit) | (hi << (64 - bit)));
        const u32 window = carry + ((u32)buf & mask);
        if ((window & 1) == 0) {
            pos += 1;
            continue;
        }
        if (window < width / 2) {
            carry = 0;
            naf[pos * BLOCK + t] = (i8)window;
        } else {
            carry = 1;
            naf[pos * BLOCK + t] = (i8)((int)window - (int)width);
        }
        pos += w;
    }
}

// The same width-5 NAF packed for LDS: a non-zero digit is followed by at least four zeros, so any four
// consecutive positions hold at most one.  One byte per block of four positions:
//   bit 7 = the block has a digit, bits 5:4 = its position in the block, bit 3 = sign, bits 2:0 = (|d| - 1) / 2.
// 64 bytes per lane instead of 256 (16 KiB per workgroup: the NAF kernel's occupancy is then set by VGPRs).
CH_DEV void recode_wnaf5_packed(const u64 k[4], u8* naf4 /* [64][BLOCK] */, int t) {
    for (int i = 0; i < 64; i++) naf4[i * BLOCK + t] = 0;
    const u32 width = 32, mask = 31;
    u32 carry = 0;
    int pos = 0;
    while (pos < 256) {
        const int idx = pos >> 6, bit = pos & 63;
        u64 lo = idx == 0 ? k[0] : idx == 1 ? k[1] : idx == 2 ? k[2] : k[3];
        u64 hi = idx == 0 ? k[1] : idx == 1 ? k[2] : idx == 2 ? k[3] : 0ull;
        u64 buf = (bit < 64 - 5) ? (lo >> bit) : ((lo >> bit) | (hi << (64 - bit)));
        const u32 window = carry + ((u32)buf & mask);
        if ((window & 1) == 0) {
            pos += 1;
            continue;
        }
        int d;
        if (window < width / 2) {
            carry = 0;
            d = (int)window;
        } else {
            carry = 1;
            d = (int)window - (int)width;
        }
        const u32 mag = (u32)(d < 0 ? -d : d);
        naf4[(pos >> 2) * BLOCK + t] = (u8)(0x80u | ((u32)(pos & 3) << 4) | (d < 0 ? 8u : 0u) | ((mag - 1) >> 1));
        pos += 5;
    }
}
CH_DEV int naf4_digit(const u8* naf4, int pos, int t) {
    const u32 b = naf4[(pos >> 2) * BLOCK + t];
    if (!(b & 0x80u) || ((b >> 4) & 3u) != (u32)(pos & 3)) return 0;
    const int mag = (int)((b & 7u) * 2 + 1);
    return (b & 8u) ? -mag : mag;
}

CH_DEV void load_scalar(const u8* scalars, size_t i, u64 k[4]) {
    const u64* p = reinterpret_cast<const u64*>(scalars + 32 * i);
#pragma unroll
    for (int j = 0; j < 4; j++) k[j] = p[j];
}

// ---------------------------------------------------------------------------
// Fixed base, constant time (config 2): BasePointTable::multiply,
// src/lookup.rs:189-206.  The 24 KiB table sits in LDS; a lookup goes over all 8
// entries of the digit's sub-table (every lane the same addresses: an LDS
// broadcast) and each lane keeps the one it wants -- the reference's
// constant-time scan (src/lookup.rs:81-102).
// Left to the compiler, the scan-and-select form of such a lookup (read entry j, keep it where |d| == j) becomes one
// `s_cbranch_execz` per read -- a branch on the digits the wave holds -- around single LDS reads that are each waited
// for.  It is written out instead: entry j is read once per wave at its wave-uniform address with only the lanes that
// want it enabled (EXEC = {|d| == j}), straight into the result registers.  No select instructions, no branch and no
// address that depends on a digit, and the N x 6 reads of a lookup are all in flight before the one wait.
// GENERIC: the table may hold identities (a caller-supplied base of small order), stored as (0, 1).
template <int J>
CH_DEV void lds_read_entry_where(u64x2 (&q)[6], u32 addr, u32 key) {
    u64 saved;
    asm volatile(
        "v_cmp_eq_u32_e32 vcc, %[j], %[key]\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "ds_read_b128 %[q0], %[a] offset:%[o0]\n\t"
        "ds_read_b128 %[q1], %[a] offset:%[o1]\n\t"
        "ds_read_b128 %[q2], %[a] offset:%[o2]\n\t"
        "ds_read_b128 %[q3], %[a] offset:%[o3]\n\t"
        "ds_read_b128 %[q4], %[a] offset:%[o4]\n\t"
        "ds_read_b128 %[q5], %[a] offset:%[o5]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]), [q4] "+v"(q[4]), [q5] "+v"(q[5]), [sv] "=&s"(saved)
        : [key] "v"(key), [a] "v"(addr), [j] "n"(J), [o0] "n"((J - 1) * 96), [o1] "n"((J - 1) * 96 + 16),
          [o2] "n"((J - 1) * 96 + 32), [o3] "n"((J - 1) * 96 + 48), [o4] "n"((J - 1) * 96 + 64), [o5] "n"((J - 1) * 96 + 80)
        : "vcc", "memory");
}
template <bool GENERIC, int... Js>
CH_DEV Affine lds_lookup_seq(const u64* sub, int d, std::integer_sequence<int, Js...>) {
    const int neg = d < 0;
    const u32 dabs = (u32)(neg ? -d : d);  // -128 -> 128: entry 128, negated (src/lookup.rs:86-98)
    const u32 addr = (u32)(uintptr_t)(const __attribute__((address_space(3))) u64*)sub;
    u64x2 q[6];
#pragma unroll
    for (int l = 0; l < 6; l++) q[l] = u64x2{0, 0};
    q[3].x = 1;  // (0, 1): what a lane with digit 0 keeps
    (lds_read_entry_where<Js + 1>(q, addr, dabs), ...);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]));
    Affine t;
#pragma unroll
    for (int l = 0; l < 3; l++) {
        t.x.c[2 * l] = q[l].x;
        t.x.c[2 * l + 1] = q[l].y;
        t.y.c[2 * l] = q[3 + l].x;
        t.y.c[2 * l + 1] = q[3 + l].y;
    }
    t.inf = dabs == 0;
    if (GENERIC) t.inf = t.inf | affine_is_identity_encoding(t.x, t.y);  // `|`: no short-circuit branch on the digit
    return affine_cond_neg(t, neg);
}
// entry |d| (1-based) of N affine entries of 96 bytes at `sub` (LDS, 16-byte aligned), negated for d < 0; (0, 1, inf) for d = 0
template <int N, bool GENERIC = false>
CH_DEV Affine lds_lookup(const u64* sub, int d) {
    return lds_lookup_seq<GENERIC>(sub, d, std::make_integer_sequence<int, N>{});
}

// A scalar outside what BasePointTable's LookupTable<8> covers (top nibble > 7): the element is
// rejected, whatever the output form: limbs zeroed, inf = 0xFF.
CH_DEV void store_rejected(u64* out, u8* inf, size_t i, int width) {
    for (int l = 0; l < width; l++) out[(size_t)width * i + l] = 0;
    inf[i] = 0xFF;
}

template <int OUT_FORM, bool GENERIC = false>
__global__ void __launch_bounds__(BLOCK, CH_GEN_WAVES) k_mul_by_generator(const DeviceTables* __restrict__ tab,
                                                            const u64* __restrict__ comb, const u8* __restrict__ scalars,
                                                            u64* __restrict__ out, u8* __restrict__ inf, size_t n) {
    __shared__ __align__(16) u64 s_tab[32 * 8 * 12];
    __shared__ i8 s_digits[64 * BLOCK];
    const int t = threadIdx.x;
    for (int w = t; w < 32 * 8 * 12; w += BLOCK) s_tab[w] = comb[w];
    const Affine shift = load_affine(tab->shift);
    const Affine mshift4 = load_affine(tab->minus_shift_4);
    const size_t nround = (n + BLOCK - 1) / BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t i = round * BLOCK + t;
        const bool active = i < n;
        u64 k[4];
        load_scalar(scalars, active ? i : n - 1, k);
        __syncthreads();
        recode_radix16(k, s_digits, t);
        __syncthreads();
        const bool in_range = (k[3] >> 60) <= 7;  // LookupTable<8> covers the top digit only then (Q3)
        Proj acc = proj_from_affine(shift);
        // the odd digits, four doublings, the even digits (src/lookup.rs:197-204): one loop body for the 64 additions,
        // its field products inlined
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
#pragma unroll 1
            for (int jj = 0; jj < 32; jj++)
                acc = proj_add_mixed_unchecked<true>(acc, lds_lookup<8, GENERIC>(s_tab + jj * 96, s_digits[(2 * jj + 1 - pass) * BLOCK + t]));
            if (pass == 0)
                for (int r = 0; r < 4; r++) acc = proj_double_unchecked(acc);
        }
        acc = proj_add_mixed_unchecked(acc, mshift4);
        if (active) {
            if (in_range)
                emit_point<OUT_FORM>(out, inf, i, acc);
            else
                store_rejected(out, inf, i, OUT_FORM == 0 ? 12 : 18);
        }
    }
}

// ---------------------------------------------------------------------------
// Fixed base, constant time, AFFINE results: the same point as BasePointTable::multiply
// (src/lookup.rs:189-206) by a schedule the hardware likes better.  An affine result does not
// depend on how k G was reached, so the comb is widened until no doubling is left: signed
// radix-64 digits d_j in [-32, 32) (k = sum d_j 64^j, 43 windows) and one 32-entry table
// (e + 1) 64^j G per window -- 43 mixed additions instead of 65 additions + 4 doublings.  The
// 129 KiB table is staged once per 1 024-thread workgroup in LDS (gfx950: 160 KiB per CU); a
// lookup goes over all 32 entries of the window at wave-uniform addresses, each lane enabled for
// the read of the entry it wants (lds_lookup; constant time, as the reference's scan,
// src/lookup.rs:81-102).  The accumulator is kept in XYZZ coordinates (8 M + 2 S per mixed addition,
// curve.hip.h), starts from the shift point S and has S subtracted at the end, so the incomplete
// addition never meets the identity (the reference's own device, src/lookup.rs:196, :205).  Digits
// are recoded on the fly.
template <int IN_FORM>
__global__ void __launch_bounds__(64) k_build_wide_table(u64* wide /* [43][32][12] */, const u64* base_pt) {
    const int j = blockIdx.x, e = threadIdx.x;  // entry (e + 1) 64^j P
    if (e >= WIDE_ENTRIES) return;
    Proj base = load_input_point<IN_FORM>(base_pt, 0);
    for (int i = 0; i < WIDE_BITS * j; i++) base = proj_double(base);
    Proj acc = proj_identity();
    for (int b = 5; b >= 0; b--) {  // (e + 1) <= 32 by left-to-right double-and-add, complete formulas
        acc = proj_double(acc);
        Proj sum = proj_add(acc, base);
        if (((e + 1) >> b) & 1) acc = sum;
    }
    store_affine(wide, nullptr, (size_t)(j * WIDE_ENTRIES + e), affine_from_proj(acc));
}

template <bool GENERIC = false>
__global__ void __launch_bounds__(WIDE_BLOCK, 4) k_mul_by_generator_wide(const DeviceTables* __restrict__ tab,
                                                                        const u64* __restrict__ wide,
                                                                        const u8* __restrict__ scalars, u64* __restrict__ out,
                                                                        u8* __restrict__ inf, size_t n) {
    __shared__ __align__(16) u64 s_wide[WIDE_WINDOWS * WIDE_ENTRIES * 12];
    const int t = threadIdx.x;
    for (int w = t; w < WIDE_WINDOWS * WIDE_ENTRIES * 12; w += WIDE_BLOCK) s_wide[w] = wide[w];
    const Affine shift = load_affine(tab->shift);
    __syncthreads();
    const size_t nround = (n + WIDE_BLOCK - 1) / WIDE_BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t i = round * WIDE_BLOCK + t;
        const bool active = i < n;
        u64 k[4];
        load_scalar(scalars, active ? i : n - 1, k);
        const bool in_range = (k[3] >> 60) <= 7;  // the range BasePointTable::multiply covers (Q3); flagged as there
        XYZZ acc = xyzz_from_affine(shift);
        int carry = 0;
#pragma unroll 1
        for (int j = 0; j < WIDE_WINDOWS; j++) {
            int d = (int)(k[0] & 63) + carry;  // the top window holds bits 252..255 only: never re-centred
            carry = d >= 32 ? 1 : 0;           // d in [0, 64]
            d -= carry << 6;
            k[0] = (k[0] >> 6) | (k[1] << 58);
            k[1] = (k[1] >> 6) | (k[2] << 58);
            k[2] = (k[2] >> 6) | (k[3] << 58);
            k[3] >>= 6;
            acc = xyzz_add_mixed_unchecked<true>(acc, lds_lookup<WIDE_ENTRIES, GENERIC>(s_wide + j * (WIDE_ENTRIES * 12), d));  // products inlined
        }
        acc = xyzz_add_mixed_unchecked(acc, affine_cond_neg(shift, true));
        if (active) {
            if (in_range)
                store_affine(out, inf, i, affine_from_xyzz(acc));
            else
                store_rejected(out, inf, i, 12);
        }
    }
}

// ---------------------------------------------------------------------------
// Fixed base through BASEPOINT_LOOKUP (`&BASEPOINT_LOOKUP * &Scalar`,
// src/constants.rs:102-135): radix-256 signed digits (src/scalar.rs:555-576), a
// modified-Jacobian accumulator, 32 x (8 doublings + constant-time scan of the
// 128-entry table + mixed addition).  3.1x the field multiplications of the
// BASEPOINT_TABLE comb; kept because it is the reference's Jacobian fixed-base
// operator and returns its (X, Y, Z).
CH_DEV void recode_radix256(const u64 k[4], i8* digits /* [32][BLOCK] */, int t) {
    int carry = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        u64 v = k[w];
        for (int j = 0; j < 8; j++) {
            const int i = 8 * w + j;
            int d = (int)(v & 0xff) + carry;
            v >>= 8;
            if (i < 31) {
                carry = (d + 128) >> 8;
                d -= carry << 8;
            }
            digits[i * BLOCK + t] = (i8)d;  // digit 31 wraps like the reference's `as i8` (128 -> -128)
        }
    }
}

template <int OUT_FORM>
__global__ void __launch_bounds__(BLOCK, CH_GEN_WAVES) k_mul_by_generator_lookup(const DeviceTables* __restrict__ tab,
                                                                   const u8* __restrict__ scalars, u64* __restrict__ out,
                                                                   u8* __restrict__ inf, size_t n) {
    __shared__ __align__(16) u64 s_tab[128 * 12];
    __shared__ i8 s_digits[32 * BLOCK];
    const int t = threadIdx.x;
    for (int w = t; w < 128 * 12; w += BLOCK) s_tab[w] = tab->basepoint_lookup[w];
    const Affine shift = load_affine(tab->shift);
    const Affine mshift = load_affine(tab->minus_shift_256);
    const size_t nround = (n + BLOCK - 1) / BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t i = round * BLOCK + t;
        const bool active = i < n;
        u64 k[4];
        load_scalar(scalars, active ? i : n - 1, k);
        __syncthreads();
        recode_radix256(k, s_digits, t);
        __syncthreads();
        MJac acc = mjac_from_affine(shift);
        for (int j = 31; j >= 0; j--) {
            acc = Ladder<MJac>::dbls_then_add(acc, 8);
            acc = Ladder<MJac>::add_after_dbl(acc, lds_lookup<128>(s_tab, s_digits[j * BLOCK + t]));
        }
        acc = Ladder<MJac>::add_last(acc, mshift);
        if (active) emit_point<OUT_FORM>(out, inf, i, acc);
    }
}

// ---------------------------------------------------------------------------
// Variable base (configs 3 and 5).  Each lane owns a window table in global
// scratch, laid out [entry][limb][lane] per workgroup so that every access is a
// coalesced 512 B per wave.
struct LaneScratch {
    // per workgroup: tab[16][12][BLOCK], z[16][6][BLOCK], prefix[16][6][BLOCK]
    static constexpr size_t WORDS_PER_BLOCK = (size_t)16 * 24 * BLOCK;
};

CH_DEV void sc_store(u64* base, int entry, int limb0, const Fp6& v, int t, int rows) {
#pragma unroll
    for (int l = 0; l < 6; l++) base[((size_t)entry * rows + limb0 + l) * BLOCK + t] = v.c[l];
}
CH_DEV Fp6 sc_load(const u64* base, int entry, int limb0, int t, int rows) {
    Fp6 v;
#pragma unroll
    for (int l = 0; l < 6; l++) v.c[l] = base[((size_t)entry * rows + limb0 + l) * BLOCK + t];
    return v;
}

// Builds the affine multiples m_0 P, m_1 P, ... in the lane's scratch table:
//   STEP2 = false: P, 2P, ..., NP         (LookupTable<N>,    src/lookup.rs:25-37)
//   STEP2 = true : P, 3P, ..., (2N-1)P    (NafLookupTable<N>, src/naf_lookup.rs:35-48)
// with the reference's complete additions and one Montgomery inversion per
// lane (its batch_normalize, src/curve/projective.rs:897-930).  Returns a bit
// mask of the entries that are the identity.
template <bool STEP2>
CH_DEV u32 build_table(const Proj& p, int n_entries, u64* s_tab, u64* s_z, u64* s_prefix, int t) {
    Proj step = STEP2 ? proj_double(p) : p;
    Proj cur = p;
    Fp6 acc = fp6_one();
    u32 ident = 0;
    for (int j = 0; j < n_entries; j++) {
        if (j > 0) cur = proj_add(step, cur);
        sc_store(s_tab, j, 0, cur.x, t, 12);
        sc_store(s_tab, j, 6, cur.y, t, 12);
        sc_store(s_z, j, 0, cur.z, t, 6);
        sc_store(s_prefix, j, 0, acc, t, 6);
        const bool skip = fp6_is_zero(cur.z);
        ident |= skip ? (1u << j) : 0u;
        Fp6 m = fp6_mul(acc, cur.z);
        acc = fp6_select(m, acc, skip);
    }
    bool some;
    acc = fp6_invert(acc, &some);
    for (int j = n_entries - 1; j >= 0; j--) {
        Fp6 z = sc_load(s_z, j, 0, t, 6);
        Fp6 zi = fp6_mul(sc_load(s_prefix, j, 0, t, 6), acc);
        const bool skip = (ident >> j) & 1;
        Fp6 m = fp6_mul(acc, z);
        acc = fp6_select(m, acc, skip);
        Fp6 x = fp6_mul(sc_load(s_tab, j, 0, t, 12), zi);
        Fp6 y = fp6_mul(sc_load(s_tab, j, 6, t, 12), zi);
        sc_store(s_tab, j, 0, fp6_select(x, fp6_zero(), skip), t, 12);
        sc_store(s_tab, j, 6, fp6_select(y, fp6_one(), skip), t, 12);
    }
    return ident;
}

// The same table [P, 2P, ..., NP] (LookupTable<N>, N = 8 or 16) for an AFFINE P by affine doublings and additions
// whose inversions are shared level by level (Montgomery's trick over the 1, 2, 4(, 8) operations of a level):
//   level (lo, 2 lo]:  m even: m P = 2 (m/2 P);   m odd: m P = ((m - 1)/2) P + ((m + 1)/2) P
// 2 M + 1..2 S per entry + 3 inversions (4 for N = 16) instead of N - 1 complete additions (12 M each) and a
// normalisation -- the entries are affine points, so they do not depend on how they were reached.  Every exceptional
// case (P the identity, or of order <= N: a doubling of a 2-torsion point, an addition of equal or opposite points)
// makes a denominator zero, hence the level's product zero and its inversion `none`: the function then returns false
// and the caller rebuilds that wave's tables with the reference's complete formulas (build_table).
// One more case has to go there although no denominator vanishes (quirk Q7): the reference fills its table with the
// Renes-Costello-Batina addition, which on THIS curve (even cofactor) has an exceptional pair whenever the difference of
// its operands is the point of order 2 -- it then returns (0, 0, 0), stored as the identity.  In the chain
// (i + 1) P = P + i P that happens iff some multiple m P, m < N, has y = 0 (P of order 2 or 10: the only even orders
// below 2 N that divide the cofactor), so a multiple with y = 0 anywhere in the table sends the wave to the reference's
// construction, which reproduces what the reference returns for such a point.
CH_DEV bool build_table_affine(const Affine& P, int n_entries, u64* s_tab, u64* s_z, u64* s_prefix, int t) {
    bool bad = P.inf | fp6_is_zero(P.y);
    sc_store(s_tab, 0, 0, P.x, t, 12);
    sc_store(s_tab, 0, 6, P.y, t, 12);
    for (int lo = 1; lo < n_entries; lo *= 2) {
        Fp6 acc = fp6_one();
        for (int i = 0; i < lo; i++) {  // denominators d_i of entry m = lo + 1 + i, and their running products
            const int m = lo + 1 + i;
            Fp6 d;
            if (m & 1)
                d = fp6_sub(sc_load(s_tab, (m + 1) / 2 - 1, 0, t, 12), sc_load(s_tab, (m - 1) / 2 - 1, 0, t, 12));  // x_b - x_a
            else
                d = fp6_double(sc_load(s_tab, m / 2 - 1, 6, t, 12));  // 2 y
            sc_store(s_z, i, 0, d, t, 6);
            if (i == 0) {
                acc = d;
            } else {
                sc_store(s_prefix, i, 0, acc, t, 6);
                acc = fp6_mul(acc, d);
            }
        }
        bool some;
        acc = fp6_invert(acc, &some);
        bad = bad || !some;
        for (int i = lo - 1; i >= 0; i--) {
            const int m = lo + 1 + i;
            Fp6 inv = acc;  // 1 / d_i
            if (i > 0) {
                inv = fp6_mul(sc_load(s_prefix, i, 0, t, 6), acc);
                acc = fp6_mul(acc, sc_load(s_z, i, 0, t, 6));
            }
            Fp6 x3, y3;
            if (m & 1) {
                const int a = (m - 1) / 2 - 1, b = (m + 1) / 2 - 1;
                const Fp6 xa = sc_load(s_tab, a, 0, t, 12), ya = sc_load(s_tab, a, 6, t, 12);
                const Fp6 lam = fp6_mul(fp6_sub(sc_load(s_tab, b, 6, t, 12), ya), inv);
                x3 = fp6_sub(fp6_sub(fp6_square(lam), xa), sc_load(s_tab, b, 0, t, 12));
                y3 = fp6_sub(fp6_mul(lam, fp6_sub(xa, x3)), ya);
            } else {
                const int h = m / 2 - 1;
                const Fp6 xh = sc_load(s_tab, h, 0, t, 12);
                Fp6 num = fp6_triple(fp6_square(xh));
                num.c[0] = fp_add(num.c[0], 1ull);  // 3 x^2 + a, a = 1
                const Fp6 lam = fp6_mul(num, inv);
                x3 = fp6_sub(fp6_square(lam), fp6_double(xh));
                y3 = fp6_sub(fp6_mul(lam, fp6_sub(xh, x3)), sc_load(s_tab, h, 6, t, 12));
            }
            sc_store(s_tab, m - 1, 0, x3, t, 12);
            sc_store(s_tab, m - 1, 6, y3, t, 12);
            bad = bad | fp6_is_zero(y3);
        }
    }
    return !bad;
}

// LookupTable<N>::from for the lane's input point: the shared-inversion affine construction, and for a wave that holds
// an exceptional point the reference's own construction.  Returns the identity mask of the entries.
// wave_unfit (the vartime fast paths): set instead of rebuilding when the wave holds a point the affine construction
// refuses (the identity in any encoding, small order, quirk Q7, off the curve) -- for such a point the reference's
// multiply_vartime is its NAF ladder over its NafLookupTable, which no LookupTable reproduces (a ProjectivePoint
// (X, Y, 0) with X != 0, for one, is multiplied into arbitrary values by multiply() and into the identity by
// multiply_vartime()), so the caller leaves the wave to the NAF kernel.
template <int IN_FORM>
CH_DEV u32 build_lookup_table(const u64* pts, size_t i, int n_entries, u64* s_tab, u64* s_z, u64* s_prefix, int t,
                              bool reference_only = false, bool* wave_unfit = nullptr) {
    Affine pa;
    if (IN_FORM == 1) {
        pa = affine_from_proj(load_proj(pts + 18 * i));
    } else {
        const Proj p = load_input_point<IN_FORM>(pts, i);  // affine already (IN_FORM 2: normalised, as the reference does)
        pa.x = p.x;
        pa.y = p.y;
        pa.inf = fp6_is_zero(p.z);
    }
    // The chord / tangent steps know nothing of B: for a point that is NOT on the curve they would fill the table with
    // its multiples on another curve, while the reference's complete additions (which use 3B) produce other values --
    // and those are what LookupTable::from returns (such inputs exist: from_uncompressed_unchecked, unchecked decoding).
    // An off-curve lane therefore sends its wave to the reference's construction as well (2 S + 1 M per lane).
    bool ok = false;
    if (!reference_only) {  // wave-uniform (a flag of the context)
        ok = pa.inf || affine_on_curve(pa.x, pa.y);
        ok = build_table_affine(pa, n_entries, s_tab, s_z, s_prefix, t) && ok;
    }
    if (__any(!ok)) {
        if (wave_unfit) {
            *wave_unfit = true;
            return 0u;
        }
        return build_table<false>(load_input_point<IN_FORM>(pts, i), n_entries, s_tab, s_z, s_prefix, t);
    }
    return 0u;
}

// NafLookupTable<8>::from (src/naf_lookup.rs:35-48): the odd multiples P, 3P, ..., 15P as affine points, by the same
// shared-inversion affine steps.  With E = 2P, 4P, 8P in turn: (2 have + 2i + 1) P = E + (2i + 1) P for the `have` odd
// multiples already there, and the next E = 2 E, one inversion per level: 4 inversions + 10 chord / tangent steps +
// Montgomery's trick (~ 42 k instructions) instead of a doubling, 7 complete additions and a normalisation (~ 64 k).
// Every exceptional case (the identity, a point of order <= 15) makes a denominator zero; the caller then rebuilds
// the wave's tables with the reference's construction, as it does for a point that is not on the curve -- and for a
// point with a multiple of y = 0 among the ones computed here (quirk Q7, see build_table_affine: the reference's chain
// (2i + 1) P = 2 P + (2i - 1) P meets its formula's exceptional pair iff (2i - 3) P is the point of order 2).
CH_DEV bool build_naf_table_affine(const Affine& P, u64* s_tab, u64* s_z, u64* s_prefix, int t) {
    bool bad = P.inf | fp6_is_zero(P.y);
    sc_store(s_tab, 0, 0, P.x, t, 12);
    sc_store(s_tab, 0, 6, P.y, t, 12);
    auto tangent = [](const Affine& e, const Fp6& inv) {  // 2 e, inv = 1 / (2 y)
        Fp6 num = fp6_triple(fp6_square(e.x));
        num.c[0] = fp_add(num.c[0], 1ull);  // 3 x^2 + a, a = 1
        const Fp6 lam = fp6_mul(num, inv);
        Affine r;
        r.x = fp6_sub(fp6_square(lam), fp6_double(e.x));
        r.y = fp6_sub(fp6_mul(lam, fp6_sub(e.x, r.x)), e.y);
        r.inf = false;
        return r;
    };
    bool some;
    Affine E = tangent(P, fp6_invert(fp6_double(P.y), &some));
    bad = bad | !some | fp6_is_zero(E.y);
    for (int have = 1; have < 8; have *= 2) {
        const int nd = have + (have < 4 ? 1 : 0);  // the chords' denominators x_E - x_i, then the tangent's 2 y_E
        Fp6 acc = fp6_one();
        for (int i = 0; i < nd; i++) {
            const Fp6 d = i < have ? fp6_sub(E.x, sc_load(s_tab, i, 0, t, 12)) : fp6_double(E.y);
            sc_store(s_z, i, 0, d, t, 6);
            if (i == 0) {
                acc = d;
            } else {
                sc_store(s_prefix, i, 0, acc, t, 6);
                acc = fp6_mul(acc, d);
            }
        }
        acc = fp6_invert(acc, &some);
        bad = bad || !some;
        Affine next = E;
        for (int i = nd - 1; i >= 0; i--) {
            Fp6 inv = acc;  // 1 / d_i
            if (i > 0) {
                inv = fp6_mul(sc_load(s_prefix, i, 0, t, 6), acc);
                acc = fp6_mul(acc, sc_load(s_z, i, 0, t, 6));
            }
            if (i < have) {
                const Fp6 xa = sc_load(s_tab, i, 0, t, 12), ya = sc_load(s_tab, i, 6, t, 12);
                const Fp6 lam = fp6_mul(fp6_sub(E.y, ya), inv);
                const Fp6 x3 = fp6_sub(fp6_sub(fp6_square(lam), xa), E.x);
                const Fp6 y3 = fp6_sub(fp6_mul(lam, fp6_sub(xa, x3)), ya);
                sc_store(s_tab, have + i, 0, x3, t, 12);
                sc_store(s_tab, have + i, 6, y3, t, 12);
                bad = bad | fp6_is_zero(y3);
            } else {
                next = tangent(E, inv);
            }
        }
        E = next;
        bad = bad | fp6_is_zero(E.y);
    }
    return !bad;
}

// NafLookupTable<8>::from for the lane's input point; returns the identity mask of the entries (as build_table).
template <int IN_FORM>
CH_DEV u32 build_naf_lookup_table(const u64* pts, size_t i, u64* s_tab, u64* s_z, u64* s_prefix, int t) {
#if CH_NAF_TABLE_AFFINE
    Affine pa;
    if (IN_FORM == 1) {
        pa = affine_from_proj(load_proj(pts + 18 * i));
    } else {
        const Proj p = load_input_point<IN_FORM>(pts, i);
        pa.x = p.x;
        pa.y = p.y;
        pa.inf = fp6_is_zero(p.z);
    }
    bool ok = !pa.inf && affine_on_curve(pa.x, pa.y);
    ok = build_naf_table_affine(pa, s_tab, s_z, s_prefix, t) && ok;
    if (!__any(!ok)) return 0u;
#endif
    return build_table<true>(load_input_point<IN_FORM>(pts, i), 8, s_tab, s_z, s_prefix, t);
}

// Constant-time lookup (src/lookup.rs:81-102): read every entry, keep entry |d|.
CH_DEV Affine scan_lookup(const u64* s_tab, int n_entries, u32 ident, int d, int t) {
    const int neg = d < 0;
    const int dabs = neg ? -d : d;
    Affine r = affine_identity();
    for (int j = 1; j <= n_entries; j++) {
        const bool hit = (dabs == j);
        Fp6 x = sc_load(s_tab, j - 1, 0, t, 12);
        Fp6 y = sc_load(s_tab, j - 1, 6, t, 12);
        r.x = fp6_select(r.x, x, hit);
        r.y = fp6_select(r.y, y, hit);
        r.inf = hit ? (bool)((ident >> (j - 1)) & 1) : r.inf;
    }
    return affine_cond_neg(r, neg);
}

// Variable-time lookup (LookupTable::get_point_vartime, src/lookup.rs:107-126): entry |d| by index.
CH_DEV Affine index_lookup(const u64* s_tab, u32 ident, int d, int t) {
    const int neg = d < 0;
    const int dabs = neg ? -d : d;
    Affine r = affine_identity();
    if (dabs != 0) {
        r.x = sc_load(s_tab, dabs - 1, 0, t, 12);
        r.y = sc_load(s_tab, dabs - 1, 6, t, 12);
        r.inf = (ident >> (dabs - 1)) & 1;
    }
    return affine_cond_neg(r, neg);
}

// The radix-16 ladder of multiply (src/curve/projective.rs:617-624, src/curve/jacobian.rs:603-610): from S,
// 64 x (four doublings, add the window's table entry), then - 2^256 S.  Window 63's four doublings of S
// are the constant shift16.  DIRECT: look the entry up by index (variable time) instead of scanning.
template <class ACC, bool DIRECT>
CH_DEV ACC ladder_radix16(const DeviceTables* tab, const u64* s_tab, int n_entries, u32 ident, const i8* s_digits, int t,
                          const Affine& mshift) {
    int d = s_digits[63 * BLOCK + t];
    ACC acc = Ladder<ACC>::add(Ladder<ACC>::shift16(tab), DIRECT ? index_lookup(s_tab, ident, d, t) : scan_lookup(s_tab, n_entries, ident, d, t));
    for (int j = 62; j >= 0; j--) {
        acc = Ladder<ACC>::dbls_then_add(acc, 4);
        d = s_digits[j * BLOCK + t];
        acc = Ladder<ACC>::add_after_dbl(acc, DIRECT ? index_lookup(s_tab, ident, d, t) : scan_lookup(s_tab, n_entries, ident, d, t));
    }
    return Ladder<ACC>::add_last(acc, mshift);
}

// ProjectivePoint::multiply (ACC = Proj, src/curve/projective.rs:614-625) and
// JacobianPoint::multiply / AffinePoint::multiply (ACC = MJac,
// src/curve/jacobian.rs:599-611, src/curve/affine.rs:373-375): same loop, the
// accumulator type is the only difference.
// DIRECT (variable time, affine results only): the same fixed-window ladder with the table entry
// read by index instead of scanned -- the fast path of the multiply_vartime entry point.  An affine
// result does not depend on the ladder, and for a scalar below 2^255 (every canonical one) the NAF
// ladder and this one give the same point; a wave with a scalar that has bit 255 set is left to
// k_multiply_vartime<..., ONLY_WIDE>, which runs the reference's NAF ladder for it.
template <class ACC, int IN_FORM, int OUT_FORM, bool DIRECT = false>
__global__ void __launch_bounds__(BLOCK, CH_CT_WAVES) k_multiply_ct(const DeviceTables* __restrict__ tab, const u64* __restrict__ pts,
                                                       const u8* __restrict__ scalars, u64* __restrict__ out,
                                                       u8* __restrict__ inf, size_t n, u64* __restrict__ scratch,
                                                       u8* __restrict__ redo_flags = nullptr /* DIRECT: one byte per wave of 64 items */) {
    __shared__ i8 s_digits[64 * BLOCK];
    const int t = threadIdx.x;
    u64* s_tab = scratch + (size_t)blockIdx.x * LaneScratch::WORDS_PER_BLOCK;
    u64* s_z = s_tab + (size_t)16 * 12 * BLOCK;
    u64* s_prefix = s_z + (size_t)16 * 6 * BLOCK;
    const Affine mshift = load_affine(tab->minus_shift_256);
    const size_t nround = (n + BLOCK - 1) / BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t i = round * BLOCK + t;
        const bool active = i < n;
        const size_t ii = active ? i : n - 1;
        u64 k[4];
        load_scalar(scalars, ii, k);
        __syncthreads();
        recode_radix16(k, s_digits, t);
        __syncthreads();
        // canonical scalars (< 2^255) only ever select entries 1..8; a wave builds
        // all 16 entries of LookupTable<16> only if one of its lanes needs them
        const bool wide = (k[3] >> 60) > 7;
        const int n_entries = __any(wide) ? 16 : 8;
        u32 ident;
        if (DIRECT) {
            // A wave that holds a scalar with bit 255 set (the NAF recoding truncates there), or a point the affine table
            // construction refuses (the identity, small order, off the curve, quirk Q7: for those the reference's answer
            // depends on ITS schedule and ITS table), belongs to the NAF kernel: recorded in redo_flags, which
            // k_multiply_vartime<..., ONLY_WIDE> reads.
            bool unfit = __any((int)(k[3] >> 63));
            ident = 0;
            if (!unfit) ident = build_lookup_table<IN_FORM>(pts, ii, n_entries, s_tab, s_z, s_prefix, t, false, &unfit);
            const size_t first = round * BLOCK + (size_t)(t & ~63);  // the wave's first item
            if ((t & 63) == 0 && redo_flags != nullptr && first < n) redo_flags[first >> 6] = unfit ? 1 : 0;
            if (unfit) continue;
        } else {
            ident = build_lookup_table<IN_FORM>(pts, ii, n_entries, s_tab, s_z, s_prefix, t, (tab->flags & 1) != 0);
        }
        const ACC acc = ladder_radix16<ACC, DIRECT>(tab, s_tab, n_entries, ident, s_digits, t, mshift);
        if (active) emit_point<OUT_FORM>(out, inf, i, acc);
    }
}

// ProjectivePoint::multiply_vartime (ACC = Proj, src/curve/projective.rs:633-662)
// and JacobianPoint / AffinePoint::multiply_vartime (ACC = MJac,
// src/curve/jacobian.rs:619-649, src/curve/affine.rs:384-386), with their
// quirk Q1: the "skip unset digits" loop never moves i off 255, so there are
// always 256 doublings and the correction is MINUS_SHIFT_POINT_ARRAY[256].
// ONLY_WIDE: run only the waves that hold a scalar with bit 255 set (whose NAF recoding drops the
// final carry, src/scalar.rs:585-643); the others were done by k_multiply_ct<..., DIRECT>.
template <class ACC, int IN_FORM, int OUT_FORM, bool ONLY_WIDE = false>
__global__ void __launch_bounds__(BLOCK, ONLY_WIDE ? 2 : CH_VT_WAVES) k_multiply_vartime(const DeviceTables* __restrict__ tab,
                                                            const u64* __restrict__ pts, const u8* __restrict__ scalars,
                                                            u64* __restrict__ out, u8* __restrict__ inf, size_t n,
                                                            u64* __restrict__ scratch, int uniform_scalar,
                                                            const u8* __restrict__ redo_flags = nullptr /* ONLY_WIDE */) {
    __shared__ u8 s_naf[64 * BLOCK];
    const int t = threadIdx.x;
    u64* s_tab = scratch + (size_t)blockIdx.x * LaneScratch::WORDS_PER_BLOCK;
    u64* s_z = s_tab + (size_t)16 * 12 * BLOCK;
    u64* s_prefix = s_z + (size_t)16 * 6 * BLOCK;
    const Affine shift = load_affine(tab->shift);
    const Affine mshift = load_affine(tab->minus_shift_256);
    const size_t nround = (n + BLOCK - 1) / BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t i = round * BLOCK + t;
        const bool active = i < n;
        const size_t ii = active ? i : n - 1;
        u64 k[4];
        load_scalar(scalars, uniform_scalar ? 0 : ii, k);  // uniform: is_torsion_free / clear_cofactor (one fixed scalar)
        int redo = 0;  // ONLY_WIDE: the waves k_multiply_ct<..., DIRECT> left out, as it recorded them
        if (ONLY_WIDE) {
            const size_t first = round * BLOCK + (size_t)(t & ~63);
            redo = first < n ? (int)redo_flags[first >> 6] : 0;
        }
        if (ONLY_WIDE && !__syncthreads_or(redo)) continue;  // no such wave in this workgroup
        __syncthreads();
        recode_wnaf5_packed(k, s_naf, t);
        __syncthreads();
        if (ONLY_WIDE && !__any(redo)) continue;
        const u32 ident = build_naf_lookup_table<IN_FORM>(pts, ii, s_tab, s_z, s_prefix, t);
        ACC acc = Ladder<ACC>::from_affine(shift);
        // Every lane runs ITS OWN reference sequence (256 doublings, an addition after
        // each non-zero digit), so the representative it ends with is the reference's.
        // What is scheduled is only WHEN the wave executes which step: a lane that
        // reaches a non-zero digit parks until the wave's next addition phase, and the
        // wave adds once 3/4 of it is parked (or nobody can double).  Lanes drift apart
        // by a few positions; on seeded scalars this costs ~349 doubling + ~55 addition
        // phases per wave instead of 256 + 256 for a lock-step loop.
        int pos = 255;       // next digit position of this lane
        int pd = 0;          // parked digit
        if (uniform_scalar) {
            // one scalar for the whole batch: the doublings before its first non-zero digit act on S alone, and what
            // they produce is a constant of the context (DeviceTables::shift_dbl_*, the same formulas, the same
            // representative) -- clear_cofactor's 130-bit h skips 125 of its 256 doublings
            while (pos >= 0 && naf4_digit(s_naf, pos, t) == 0) pos--;
            acc = Ladder<ACC>::shift_doubled(tab, 255 - pos);
        }
        for (;;) {
            const bool can = pd == 0 && pos >= 0;
            const u64 m_can = __builtin_amdgcn_ballot_w64(can);
            const u64 m_pend = __builtin_amdgcn_ballot_w64(pd != 0);
            if (m_can == 0 && m_pend == 0) break;
            if (m_pend != 0 && (__builtin_popcountll(m_pend) >= CH_VT_PARK || m_can == 0)) {
                if (pd != 0) {
                    const int e = (pd < 0 ? -pd : pd) >> 1;  // NafLookupTable::get_point(x) = table[x / 2]
                    Affine q;
                    q.x = sc_load(s_tab, e, 0, t, 12);
                    q.y = sc_load(s_tab, e, 6, t, 12);
                    q.inf = (ident >> e) & 1;
                    acc = Ladder<ACC>::add(acc, affine_cond_neg(q, pd < 0));
                    pd = 0;
                }
                continue;
            }
            if (can) {
                acc = Ladder<ACC>::dbl_inl(acc);
                pd = naf4_digit(s_naf, pos, t);
                pos--;
            }
        }
        acc = Ladder<ACC>::add(acc, mshift);
        if (active) emit_point<OUT_FORM>(out, inf, i, acc);
    }
}


// ---------------------------------------------------------------------------
// multiply_vartime with the reference's own representative (L2), pooled: the NAF ladder of k_multiply_vartime with
// every WAVE owning SLOTS x 64 scalar multiplications instead of 64.  Each item still runs ITS OWN reference sequence
// (256 doublings, an addition after each non-zero width-5 NAF digit, then - 2^256 S; src/curve/projective.rs:633-662,
// src/curve/jacobian.rs:619-649, quirk Q1), so the (X, Y, Z) it ends with is the reference's; what the wave schedules is
// which 64 of its items take the next step together.  With one item per lane (k_multiply_vartime) a doubling phase
// keeps ~ 73 % of the lanes busy and an addition phase ~ 78 %, because at every position some lane has a non-zero
// digit; with a pool of 128 a phase is full except at the very end of a round (tools/naf_pool_sim.py: 97 - 99 % of the
// lane-slots used, 329 phase-units per 64 items against 440; on the device 99.6 % / 99.0 % of the lanes of a doubling /
// addition phase, 257 + 43 phases per 64 items).  The price is that an item's accumulator lives in global scratch
// between its steps (AccIO: one contiguous record per item, 16-byte accesses) and is loaded / stored around every
// step: 2 x 144 (192) bytes per step and lane.
// Everything an item owns -- its NAF in LDS, its table and accumulator in scratch, its position and parked digit in
// LDS -- belongs to ONE wave, so the scheduling loop has no barrier, no lock and no wait on another wave: every phase
// completes at least one step of one item, which bounds a round by SLOTS * 64 * (256 + 52) phases (a non-zero NAF digit
// is followed by four zeros).  The loop enforces that bound and flags what is unfinished (zero limbs, inf = 0xFE)
// instead of spinning.
//   phase policy (the model in tools/naf_pool_sim.py, statement for statement):
//     additions (each with the run of doublings that follows it in the item's sequence, see CH_VT_RUN) if >= 64 items
//     are parked at a non-zero digit, else single doublings if >= 64 can double, else the larger group;
//     among the candidates those at or above the candidates' mean position first (the laggards), slots and lanes in an
//     order that rotates with the phase number.
#ifndef CH_VT_SLOTS
#define CH_VT_SLOTS 2
#endif
// Measured (tools/lab/vt_lab.py, 2^22, affine in; parked = k_multiply_vartime): projective representative 18.5 M/s parked,
// pooled with one step per phase 21.7-21.9 at 3 waves per SIMD and 2 slots (20.8 at 4 waves, 20.6 with 3 slots, 20.8 with
// the accumulators SoA like the tables, 20.9 with the doublings through calls); Jacobian representative 23.1 parked, pooled
// 24.7-24.9.  With no accumulator traffic at all that schedule would reach 24.8 / 30.5: its loads and its stores each cost
// ~ 6 % (SQ_INSTS_VALU + 9 % per phase for addresses, packing and the scheduling itself).  Hence the fused runs
// (CH_VT_RUN): an addition phase goes on with the doublings that follow it -- 24.1 / 29.4 M/s with runs of 5 that
// shorten to 4 below position 60 (23.7 / 28.7 with runs of 4 throughout, 23.0 / 28.3 with 5 throughout, 22.8 / 27.5 with
// 3 slots; 4 waves per SIMD the same as 3).
#ifndef CH_VT_POOL_WAVES
#define CH_VT_POOL_WAVES 3
#endif
// doublings fused behind an addition in one phase (0: none, the first form of this kernel), and the mean position of a
// phase's candidates below which one fewer is fused
#ifndef CH_VT_RUN
#define CH_VT_RUN 5
#endif
#ifndef CH_VT_RUN_SWITCH
#define CH_VT_RUN_SWITCH 60
#endif
#ifdef LAB_COUNT
__device__ unsigned long long lab_counters[4];  // tools/lab only: doubling phases, addition phases, lanes used in each
#endif
struct VtPool {
    static constexpr int SLOTS = CH_VT_SLOTS;
    static constexpr size_t TAB = (size_t)8 * 12 * BLOCK;   // NafLookupTable<8> per slot
    static constexpr size_t ZP = (size_t)8 * 6 * BLOCK;     // z / prefix of the table build (shared by the slots)
    static size_t words(int n_fp6) { return SLOTS * TAB + 2 * ZP + (size_t)SLOTS * n_fp6 * 6 * BLOCK; }
    static constexpr int phase_bound() { return SLOTS * 64 * (256 + 52); }
};
CH_DEV int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// rank of `lane` among the set bits of m, counted from lane r upwards (wrapping)
CH_DEV int rotated_rank(u64 m, int r, int lane) {
    const u64 rot = r ? ((m >> r) | (m << (64 - r))) : m;
    const int lr = (lane - r) & 63;
    return __builtin_popcountll(rot & ((1ull << lr) - 1ull));
}
// An item's accumulator in scratch: NF x 48 contiguous bytes per item (AoS, [slot][owner lane][limb]), moved by 16-byte
// accesses -- a phase's lanes read whole items, so the layout that coalesces is the one that keeps an item together.
// (Measured, tools/lab/vt_lab.py: the SoA form of the window tables, 8-byte gathers at [limb][lane], costs the pooled
// kernel 1.7 us of SIMD time per phase: 20.8 M/s against 24.8 with no state traffic at all.)
template <class ACC> struct AccIO;
template <> struct AccIO<Proj> {
    static constexpr int NF = 3;
    static CH_DEV void store(u64* st, const Proj& a, int tt) {
        u64* p = st + (size_t)tt * (NF * 6);
        store_fp6_vec_raw(p, a.x); store_fp6_vec_raw(p + 6, a.y); store_fp6_vec_raw(p + 12, a.z);
    }
    static CH_DEV Proj load(const u64* st, int tt) {
        const u64* p = st + (size_t)tt * (NF * 6);
        Proj a; a.x = load_fp6_vec(p); a.y = load_fp6_vec(p + 6); a.z = load_fp6_vec(p + 12); return a;
    }
};
template <> struct AccIO<MJac> {
    static constexpr int NF = 4;
    static CH_DEV void store(u64* st, const MJac& a, int tt) {
        u64* p = st + (size_t)tt * (NF * 6);
        store_fp6_vec_raw(p, a.x); store_fp6_vec_raw(p + 6, a.y); store_fp6_vec_raw(p + 12, a.z); store_fp6_vec_raw(p + 18, a.z4);
    }
    static CH_DEV MJac load(const u64* st, int tt) {
        const u64* p = st + (size_t)tt * (NF * 6);
        MJac a; a.x = load_fp6_vec(p); a.y = load_fp6_vec(p + 6); a.z = load_fp6_vec(p + 12); a.z4 = load_fp6_vec(p + 18); return a;
    }
};

template <class ACC, int IN_FORM, int OUT_FORM>
__global__ void __launch_bounds__(BLOCK, CH_VT_POOL_WAVES) k_multiply_vartime_pool(const DeviceTables* __restrict__ tab,
                                                            const u64* __restrict__ pts, const u8* __restrict__ scalars,
                                                            u64* __restrict__ out, u8* __restrict__ inf, size_t n,
                                                            u64* __restrict__ scratch, size_t words_per_block) {
    constexpr int SLOTS = VtPool::SLOTS, NF = AccIO<ACC>::NF;
    __shared__ u8 s_naf[SLOTS * 64 * BLOCK];
    __shared__ short s_pos[SLOTS * BLOCK];
    __shared__ i8 s_pd[SLOTS * BLOCK];
    __shared__ u8 s_ident[SLOTS * BLOCK];
    __shared__ u8 s_list[BLOCK];
    const int t = threadIdx.x, lane = t & 63, wbase = t & ~63;
    u64* base = scratch + (size_t)blockIdx.x * words_per_block;
    u64* s_z = base + SLOTS * VtPool::TAB;
    u64* s_prefix = s_z + VtPool::ZP;
    u64* state0 = s_prefix + VtPool::ZP;
    const Affine shift = load_affine(tab->shift);
    const Affine mshift = load_affine(tab->minus_shift_256);
    const size_t per_round = (size_t)SLOTS * BLOCK;
    const size_t nround = (n + per_round - 1) / per_round;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        // ---- set-up: lane t owns item round * SLOTS * BLOCK + s * BLOCK + t of every slot s ----
        __syncthreads();
        for (int s = 0; s < SLOTS; s++) {
            const size_t i = round * per_round + (size_t)s * BLOCK + t;
            const bool active = i < n;
            const size_t ii = active ? i : n - 1;
            u64 k[4];
            load_scalar(scalars, ii, k);
            recode_wnaf5_packed(k, s_naf + (size_t)s * 64 * BLOCK, t);
            s_pos[s * BLOCK + t] = active ? (short)255 : (short)-1;
            s_pd[s * BLOCK + t] = 0;
            if (__any(active)) {
                s_ident[s * BLOCK + t] = (u8)build_naf_lookup_table<IN_FORM>(pts, ii, base + s * VtPool::TAB, s_z, s_prefix, t);
                AccIO<ACC>::store(state0 + (size_t)s * NF * 6 * BLOCK, Ladder<ACC>::from_affine(shift), t);
            }
        }
        __threadfence_block();
        __syncthreads();
        // ---- the wave's scheduling loop: no barrier, nothing shared with another wave ----
        int phase = 0;
        for (;; phase++) {
            int pos_s[SLOTS], pd_s[SLOTS];
            u64 m_d[SLOTS], m_a[SLOTS];
            int n_d = 0, n_a = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; s++) {
                pos_s[s] = s_pos[s * BLOCK + t];
                pd_s[s] = s_pd[s * BLOCK + t];
                m_d[s] = __builtin_amdgcn_ballot_w64(pd_s[s] == 0 && pos_s[s] >= 0);
                m_a[s] = __builtin_amdgcn_ballot_w64(pd_s[s] != 0);
                n_d += __builtin_popcountll(m_d[s]);
                n_a += __builtin_popcountll(m_a[s]);
            }
            if ((n_d == 0 && n_a == 0) || phase >= VtPool::phase_bound()) break;
            const bool add = n_a >= 64 || (n_d < 64 && n_a > n_d);
            const int cnt = add ? n_a : n_d;
            // the candidates at or above their mean position go first
            int mine = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; s++) {
                const bool want = add ? pd_s[s] != 0 : (pd_s[s] == 0 && pos_s[s] >= 0);
                mine += want ? pos_s[s] : 0;
            }
            const int sum = wave_sum_i32(mine);
            u64 m_hi[SLOTS], m_lo[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; s++) {
                const bool want = add ? pd_s[s] != 0 : (pd_s[s] == 0 && pos_s[s] >= 0);
                m_hi[s] = __builtin_amdgcn_ballot_w64(want && pos_s[s] * cnt >= sum);
                m_lo[s] = (add ? m_a[s] : m_d[s]) & ~m_hi[s];
            }
            const int r = (phase * 17) & 63;
            int offset = 0;
#pragma unroll
            for (int tier = 0; tier < 2; tier++) {
#pragma unroll
                for (int o = 0; o < SLOTS; o++) {
                    const int s = (o + phase) % SLOTS;
                    const u64 m = tier == 0 ? m_hi[s] : m_lo[s];
                    if ((m >> lane) & 1ull) {
                        const int p = offset + rotated_rank(m, r, lane);
                        if (p < 64) s_list[wbase + p] = (u8)((s << 6) | lane);
                    }
                    offset += __builtin_popcountll(m);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int take = cnt < 64 ? cnt : 64;
#ifdef LAB_COUNT
            if (lane == 0) {
                atomicAdd(&lab_counters[add ? 1 : 0], 1ull);
                atomicAdd(&lab_counters[add ? 3 : 2], (unsigned long long)take);
            }
#endif
            // an addition phase goes on with the doublings that follow the addition in every item's sequence: a non-zero
            // width-5 NAF digit is followed by four zeros, so doublings 1..4 meet no digit and the fifth is the first that
            // can.  One load and one store of the accumulator for up to six steps; five while the candidates are still far
            // from the end (mean position > CH_VT_RUN_SWITCH), four after, which keeps the items in step at the tail
            // (tools/naf_pool_sim.py: 96 phases per 64 items instead of 300, 98.7 % of the lane-steps used).
            const int run = CH_VT_RUN == 0 ? 0 : (sum > CH_VT_RUN_SWITCH * cnt ? CH_VT_RUN : CH_VT_RUN - 1);
            if (lane < take) {
                const int code = s_list[wbase + lane];
                const int s = code >> 6, tt = wbase + (code & 63);
                u64* st = state0 + (size_t)s * NF * 6 * BLOCK;
                ACC acc = AccIO<ACC>::load(st, tt);
                const int p = s_pos[s * BLOCK + tt];  // the next position to double at
                int ndbl = 1;
                if (add) {
                    const int pd = s_pd[s * BLOCK + tt];
                    const int e = (pd < 0 ? -pd : pd) >> 1;  // NafLookupTable::get_point(x) = table[x / 2]
                    const u64* tb = base + s * VtPool::TAB;
                    Affine q;
                    q.x = sc_load(tb, e, 0, tt, 12);
                    q.y = sc_load(tb, e, 6, tt, 12);
                    q.inf = (s_ident[s * BLOCK + tt] >> e) & 1;
                    acc = Ladder<ACC>::add(acc, affine_cond_neg(q, pd < 0));
                    ndbl = p + 1 < run ? p + 1 : run;  // an item close to its end has fewer doublings left
                }
#pragma unroll 1
                for (int j = 0; j < (CH_VT_RUN > 1 ? CH_VT_RUN : 1); j++) {  // ONE doubling body for both kinds of phase
                    if (j >= ndbl) break;
                    acc = Ladder<ACC>::dbl_inl(acc);
                }
                // the digit the last doubling of the run arrives at (the ones before it are zeros by construction)
                s_pd[s * BLOCK + tt] = ndbl > 0 ? (i8)naf4_digit(s_naf + (size_t)s * 64 * BLOCK, p - ndbl + 1, tt) : (i8)0;
                s_pos[s * BLOCK + tt] = (short)(p - ndbl);
                AccIO<ACC>::store(st, acc, tt);
            }
            __threadfence_block();  // the accumulators just stored are read by other lanes of this wave in a later phase
            __builtin_amdgcn_wave_barrier();
        }
        // ---- the owners close their items: - 2^256 S, emit ----
        for (int s = 0; s < SLOTS; s++) {
            const size_t i = round * per_round + (size_t)s * BLOCK + t;
            if (i < n) {
                const bool finished = s_pos[s * BLOCK + t] < 0 && s_pd[s * BLOCK + t] == 0;
                ACC acc = AccIO<ACC>::load(state0 + (size_t)s * NF * 6 * BLOCK, t);
                acc = Ladder<ACC>::add(acc, mshift);
                if (finished) {
                    emit_point<OUT_FORM>(out, inf, i, acc);
                } else {  // the phase bound was hit: flagged, never silently wrong
                    const int width = OUT_FORM == 0 ? 12 : 18;
                    for (int l = 0; l < width; l++) out[(size_t)width * i + l] = 0;
                    if (inf) inf[i] = 0xFE;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Multi-scalar (Straus) ladders: out[g] = sum_{j<m} k[g][j] * P[g][j], shared
// doublings, one window table per point, additions in index order.
//   ct      ProjectivePoint / JacobianPoint::multiply_many  src/curve/projective.rs:817-832, src/curve/jacobian.rs:807-823
//           (multiply_double, :666-685 / :653-673, is the case m = 2)
//   vartime multiply_many_vartime / multiply_double_vartime  src/curve/projective.rs:693-742, :840-869
//           with_basepoint: multiply_double_with_basepoint_vartime (:756-809): one more scalar per
//           group, recoded with a width-8 NAF, multiplying G through ODD_MULTIPLES_BASEPOINT
// Per-workgroup scratch: tables [m][16][12][BLOCK], z / prefix [16][6][BLOCK] each,
// identity masks [m][BLOCK], digits [ns][256][BLOCK] bytes.
struct ManyScratch {
    static constexpr size_t TAB = (size_t)16 * 12 * BLOCK;
    static constexpr size_t ZP = (size_t)16 * 6 * BLOCK;
    static size_t words(int m, int ns) { return (size_t)m * TAB + 2 * ZP + (size_t)m * BLOCK + (size_t)ns * 256 * BLOCK / 8; }
};

// DIRECT (variable time, affine results only): table entries read by index -- the fast path of multiply_many_vartime /
// multiply_double_vartime, whose affine result is this ladder's for scalars below 2^255; a wave that holds a larger one
// is left to k_multiply_many_vartime<..., ONLY_WIDE>.
template <class ACC, int IN_FORM, int OUT_FORM, bool DIRECT = false>
__global__ void __launch_bounds__(BLOCK, CH_MANY_WAVES) k_multiply_many_ct(const DeviceTables* __restrict__ tab, const u64* __restrict__ pts,
                                                            const u8* __restrict__ scalars, u64* __restrict__ out,
                                                            u8* __restrict__ inf, size_t n_groups, int m,
                                                            u64* __restrict__ scratch, size_t words_per_block,
                                                            u8* __restrict__ redo_flags = nullptr /* DIRECT: one byte per wave of 64 groups */) {
    const int t = threadIdx.x;
    u64* s_tabs = scratch + (size_t)blockIdx.x * words_per_block;
    u64* s_z = s_tabs + (size_t)m * ManyScratch::TAB;
    u64* s_prefix = s_z + ManyScratch::ZP;
    u32* s_ident = reinterpret_cast<u32*>(s_prefix + ManyScratch::ZP);
    i8* s_dig = reinterpret_cast<i8*>(s_prefix + ManyScratch::ZP + (size_t)m * BLOCK);
    const Affine mshift = load_affine(tab->minus_shift_256);
    const size_t nround = (n_groups + BLOCK - 1) / BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t g = round * BLOCK + t;
        const bool active = g < n_groups;
        const size_t gg = active ? g : n_groups - 1;
        bool wide = false;
        for (int j = 0; j < m; j++) {
            u64 k[4];
            load_scalar(scalars, gg * m + j, k);
            recode_radix16(k, s_dig + (size_t)j * 256 * BLOCK, t);
            wide = wide || (k[3] >> 60) > 7;
        }
        const int n_entries = __any(wide) ? 16 : 8;
        if (DIRECT) {
            // top nibble > 7 <=> bit 255 set, or a point the affine table construction refuses: the wave belongs to the
            // NAF kernel (see k_multiply_ct), recorded in redo_flags
            bool unfit = __any(wide);
            for (int j = 0; j < m && !unfit; j++)
                s_ident[j * BLOCK + t] = build_lookup_table<IN_FORM>(pts, gg * m + j, n_entries, s_tabs + (size_t)j * ManyScratch::TAB, s_z,
                                                                          s_prefix, t, false, &unfit);
            const size_t first = round * BLOCK + (size_t)(t & ~63);
            if ((t & 63) == 0 && redo_flags != nullptr && first < n_groups) redo_flags[first >> 6] = unfit ? 1 : 0;
            if (unfit) continue;
        } else {
            for (int j = 0; j < m; j++)
                s_ident[j * BLOCK + t] = build_lookup_table<IN_FORM>(pts, gg * m + j, n_entries, s_tabs + (size_t)j * ManyScratch::TAB, s_z,
                                                                           s_prefix, t, (tab->flags & 1) != 0);
        }
        ACC acc = Ladder<ACC>::shift16(tab);  // window 63's four doublings of S
        for (int w = 63; w >= 0; w--) {
            if (w < 63) acc = Ladder<ACC>::dbls_then_add(acc, 4);
            for (int j = 0; j < m; j++) {
                const int d = s_dig[((size_t)j * 256 + w) * BLOCK + t];
                const u64* tb = s_tabs + (size_t)j * ManyScratch::TAB;
                const Affine q = DIRECT ? index_lookup(tb, s_ident[j * BLOCK + t], d, t) : scan_lookup(tb, n_entries, s_ident[j * BLOCK + t], d, t);
                acc = Ladder<ACC>::add_inl(acc, q);  // one loop body for the m additions of a window
            }
        }
        acc = Ladder<ACC>::add_last(acc, mshift);
        if (active) emit_point<OUT_FORM>(out, inf, g, acc);
    }
}

// ONLY_WIDE: run only the waves that hold a scalar with bit 255 set (no canonical scalar has it); the others were
// done by the composed path of multiply_double_with_basepoint_vartime (k_affine_sum below).
template <class ACC, int IN_FORM, int OUT_FORM, bool ONLY_WIDE = false>
__global__ void __launch_bounds__(BLOCK, CH_MANY_WAVES) k_multiply_many_vartime(const DeviceTables* __restrict__ tab,
                                                                 const u64* __restrict__ pts, const u8* __restrict__ scalars,
                                                                 u64* __restrict__ out, u8* __restrict__ inf, size_t n_groups,
                                                                 int m, int with_basepoint, u64* __restrict__ scratch,
                                                                 size_t words_per_block,
                                                                 const u8* __restrict__ redo_flags = nullptr /* ONLY_WIDE, optional */) {
    const int t = threadIdx.x;
    const int ns = m + (with_basepoint ? 1 : 0);
    u64* s_tabs = scratch + (size_t)blockIdx.x * words_per_block;
    u64* s_z = s_tabs + (size_t)m * ManyScratch::TAB;
    u64* s_prefix = s_z + ManyScratch::ZP;
    u32* s_ident = reinterpret_cast<u32*>(s_prefix + ManyScratch::ZP);
    i8* s_dig = reinterpret_cast<i8*>(s_prefix + ManyScratch::ZP + (size_t)m * BLOCK);
    const Affine shift = load_affine(tab->shift);
    const Affine mshift = load_affine(tab->minus_shift_256);
    const size_t nround = (n_groups + BLOCK - 1) / BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t g = round * BLOCK + t;
        const bool active = g < n_groups;
        const size_t gg = active ? g : n_groups - 1;
        if (ONLY_WIDE) {
            // the waves the fast path left out: as k_multiply_many_ct<..., DIRECT> recorded them (redo_flags), or -- for
            // the composed a P + b G path, whose a P went through the single-scalar kernels and their own fix-up -- the
            // waves with a scalar that has bit 255 set or a point off the curve
            int wide = 0;
            if (redo_flags != nullptr) {
                const size_t first = round * BLOCK + (size_t)(t & ~63);
                wide = first < n_groups ? (int)redo_flags[first >> 6] : 0;
            } else {
                for (int j = 0; j < ns; j++) wide |= (int)(scalars[(gg * ns + j) * 32 + 31] >> 7);
                for (int j = 0; j < m; j++) wide |= (int)!point_on_curve<IN_FORM>(pts, gg * m + j);
            }
            if (!__any(wide)) continue;
        }
        for (int j = 0; j < ns; j++) {
            u64 k[4];
            load_scalar(scalars, gg * ns + j, k);
            recode_wnaf(k, j < m ? 5 : 8, s_dig + (size_t)j * 256 * BLOCK, t);
        }
        for (int j = 0; j < m; j++) {
            s_ident[j * BLOCK + t] = build_naf_lookup_table<IN_FORM>(pts, gg * m + j, s_tabs + (size_t)j * ManyScratch::TAB, s_z, s_prefix, t);
        }
        ACC acc = Ladder<ACC>::from_affine(shift);
        // Same wave-level parking as k_multiply_vartime; a lane may now owe several
        // additions at one position (one per table, in index order): after each it goes
        // on scanning the remaining tables of that position before it may double again.
        int pos = 255, cur = 0, j = ns, pd = 0, pj = 0;
        for (;;) {
            const bool can = pd == 0 && j >= ns && pos >= 0;
            const u64 m_can = __builtin_amdgcn_ballot_w64(can);
            const u64 m_pend = __builtin_amdgcn_ballot_w64(pd != 0);
            if (m_can == 0 && m_pend == 0) break;
            bool rescan = false;
            if (m_pend != 0 && (__builtin_popcountll(m_pend) >= CH_VT_PARK || m_can == 0)) {
                if (pd != 0) {
                    const int e = (pd < 0 ? -pd : pd) >> 1;  // NafLookupTable::get_point(x) = table[x / 2]
                    Affine q;
                    if (pj < m) {
                        const u64* tb = s_tabs + (size_t)pj * ManyScratch::TAB;
                        q.x = sc_load(tb, e, 0, t, 12);
                        q.y = sc_load(tb, e, 6, t, 12);
                        q.inf = (s_ident[pj * BLOCK + t] >> e) & 1;
                    } else {
                        q = load_affine(tab->odd_multiples + 12 * e);
                    }
                    acc = Ladder<ACC>::add(acc, affine_cond_neg(q, pd < 0));
                    pd = 0;
                    rescan = true;
                }
            } else if (can) {
                acc = Ladder<ACC>::dbl_inl(acc);
                cur = pos;
                pos--;
                j = 0;
                rescan = true;
            }
            if (rescan) {
                while (j < ns) {
                    const int d = s_dig[((size_t)j * 256 + cur) * BLOCK + t];
                    j++;
                    if (d != 0) {
                        pd = d;
                        pj = j - 1;
                        break;
                    }
                }
            }
        }
        acc = Ladder<ACC>::add(acc, mshift);
        if (active) emit_point<OUT_FORM>(out, inf, g, acc);
    }
}


// multiply_double_with_basepoint_vartime with affine results, composed (capi.hip): the scalar pairs are split,
// a P runs through the fixed-window ladder (k_multiply_ct<..., DIRECT>), b G through the widened comb
// (k_mul_by_generator_wide), and the two affine points are added here with the complete formula and normalised.
// The sum is the affine point the reference's Straus ladder returns (src/curve/projective.rs:756-809,
// src/curve/jacobian.rs:745-799) for every pair of scalars below 2^255; waves with a larger one are redone by
// k_multiply_many_vartime<..., ONLY_WIDE>.
__global__ void __launch_bounds__(BLOCK) k_split_scalar_pairs(const u64* __restrict__ pairs, u64* __restrict__ first,
                                                              u64* __restrict__ second, size_t n) {
    for (size_t w = (size_t)blockIdx.x * BLOCK + threadIdx.x; w < n * 8; w += (size_t)gridDim.x * BLOCK) {
        const size_t i = w >> 3, l = w & 7;
        const u64 v = pairs[w];
        if (l < 4) first[4 * i + l] = v;
        else second[4 * i + l - 4] = v;
    }
}
__global__ void __launch_bounds__(BLOCK) k_affine_sum(const u64* __restrict__ a_xy, const u8* __restrict__ a_inf,
                                                      const u64* __restrict__ b_xy, const u8* __restrict__ b_inf,
                                                      u64* __restrict__ out, u8* __restrict__ inf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        Affine a = load_affine(a_xy + 12 * i), b = load_affine(b_xy + 12 * i);
        a.inf = a_inf[i] != 0;
        b.inf = b_inf[i] != 0;
        Proj r = jac_add_mixed(proj_from_affine(a), b);  // (x, y, 1) is a's Jacobian form too
        if (a.inf) r = proj_from_affine(b);              // the one case add_mixed does not select (quirk Q6)
        store_affine(out, inf, i, affine_from_jac(r));
    }
}

// ---------------------------------------------------------------------------
// Point encodings (src/curve/encoding.rs): 97-byte uncompressed and 49-byte
// compressed forms.  Records are not a multiple of 4 bytes, so a workgroup's
// 256 records are assembled in LDS and move to / from HBM as one contiguous,
// dword-coalesced span (256 * 97 and 256 * 49 are multiples of 4).
CH_DEV bool fp_lex_largest(u64 a) { return fp_canonical(a) >= 0x7fffffff80000001ull; }  // src/fp.rs:281-300
// Fp6::lexicographically_largest, src/fp6.rs:398-424: the expression exactly as written there.
CH_DEV bool fp6_lex_largest(const Fp6& a) {
    bool r = fp_lex_largest(a.c[5]);
#pragma unroll
    for (int i = 4; i >= 0; i--) r = r || (fp_is_zero(a.c[i + 1]) && fp_lex_largest(a.c[i]));
    return r;
}
CH_DEV void lds_put_fp6(u8* dst, const Fp6& a) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const u64 v = fp_canonical(a.c[i]);
#pragma unroll
        for (int b = 0; b < 8; b++) dst[8 * i + b] = (u8)(v >> (8 * b));
    }
}
CH_DEV bool lds_get_fp6(const u8* src, Fp6& a) {  // Fp6::from_bytes: none unless every limb < p (src/fp6.rs:1154-1189)
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        u64 v = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) v |= (u64)src[8 * i + b] << (8 * b);
        const bool some = v < FP_P;
        ok = ok && some;
        a.c[i] = some ? v : 0;
    }
    return ok;
}

// CompressedPoint::from_affine (REC = 49, src/curve/encoding.rs:55-77) and
// UncompressedPoint::from_affine (REC = 97, :169-183).
template <int REC>
__global__ void __launch_bounds__(BLOCK) k_encode(const u64* __restrict__ xy, const u8* __restrict__ inf, u8* __restrict__ out,
                                                  size_t n) {
    __shared__ u32 s_buf[BLOCK * REC / 4];
    u8* sb = reinterpret_cast<u8*>(s_buf);
    const int t = threadIdx.x;
    const size_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t e0 = tile * BLOCK;
        const size_t cnt = n - e0 < (size_t)BLOCK ? n - e0 : (size_t)BLOCK;
        if ((size_t)t < cnt) {
            const size_t e = e0 + t;
            const bool is_inf = inf[e] != 0;
            Fp6 x = load_fp6_vec(xy + 12 * e), y = load_fp6_vec(xy + 12 * e + 6);
            u8* rec = sb + t * REC;
            lds_put_fp6(rec, is_inf ? fp6_zero() : x);
            if (REC == 97) {
                lds_put_fp6(rec + 48, is_inf ? fp6_zero() : y);
                rec[96] = is_inf ? 0x80 : 0;
            } else {
                rec[48] = (is_inf ? 0x80 : 0) | ((!is_inf && fp6_lex_largest(y)) ? 0x40 : 0);
            }
        }
        __syncthreads();
        const size_t bytes = cnt * REC;
        u8* dst = out + e0 * REC;  // e0 * REC is a multiple of 4
        for (size_t w = t; w < bytes / 4; w += BLOCK) reinterpret_cast<u32*>(dst)[w] = s_buf[w];
        for (size_t b = (bytes / 4) * 4 + t; b < bytes; b += BLOCK) dst[b] = sb[b];
        __syncthreads();
    }
}

// UncompressedPoint::to_affine (src/curve/encoding.rs:187-226).  CHECKED adds the
// on-curve test of AffinePoint::from_uncompressed (src/curve/affine.rs:289-292, :364-368);
// its subgroup test runs afterwards as a fixed-scalar ladder that ANDs into is_some.
// A rejected record comes out as (0, 0, inf = 0) with is_some = 0.
template <int CHECKED>
__global__ void __launch_bounds__(BLOCK) k_decode_uncompressed(const u8* __restrict__ in, u64* __restrict__ xy,
                                                               u8* __restrict__ inf, u8* __restrict__ is_some, size_t n) {
    constexpr int REC = 97;
    __shared__ u32 s_buf[BLOCK * REC / 4];
    u8* sb = reinterpret_cast<u8*>(s_buf);
    const int t = threadIdx.x;
    const size_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t e0 = tile * BLOCK;
        const size_t cnt = n - e0 < (size_t)BLOCK ? n - e0 : (size_t)BLOCK;
        const size_t bytes = cnt * REC;
        const u8* src = in + e0 * REC;
        for (size_t w = t; w < bytes / 4; w += BLOCK) s_buf[w] = reinterpret_cast<const u32*>(src)[w];
        for (size_t b = (bytes / 4) * 4 + t; b < bytes; b += BLOCK) sb[b] = src[b];
        __syncthreads();
        if ((size_t)t < cnt) {
            const size_t e = e0 + t;
            const u8* rec = sb + t * REC;
            const bool flag = (rec[96] >> 7) & 1;
            Affine p;
            bool ok = lds_get_fp6(rec, p.x);
            ok = lds_get_fp6(rec + 48, p.y) && ok;
            ok = ok && (!flag || (fp6_is_zero(p.x) && fp6_is_zero(p.y)));
            p.inf = flag;
            if (flag) p = affine_identity();
            if (CHECKED && ok && !p.inf) {  // y^2 - x (x^2 + 1) == B
                Fp6 lhs = fp6_sub(fp6_square(p.y), fp6_mul(p.x, fp6_add(fp6_square(p.x), fp6_one())));
                lhs = fp6_canonical(lhs);
                ok = lhs.c[0] == 395 && lhs.c[1] == 1 && lhs.c[2] == 0 && lhs.c[3] == 0 && lhs.c[4] == 0 && lhs.c[5] == 0;
            }
            if (!ok) {
                p.x = fp6_zero();
                p.y = fp6_zero();
                p.inf = false;
            }
            store_fp6_vec_canonical(xy + 12 * e, p.x);
            store_fp6_vec_canonical(xy + 12 * e + 6, p.y);
            inf[e] = p.inf ? 1 : 0;
            is_some[e] = ok ? 1 : 0;
        }
        __syncthreads();
    }
}

// x^3 + x + B, B = u + 395 (src/curve/mod.rs:42-49): the right-hand side whose root is y.
CH_DEV Fp6 curve_rhs(const Fp6& x) {
    Fp6 r = fp6_add(fp6_mul(fp6_square(x), x), x);
    r.c[0] = fp_add(r.c[0], 395ull);
    r.c[1] = fp_add(r.c[1], 1ull);
    return r;
}

// CompressedPoint::to_affine = AffinePoint::from_compressed_unchecked (src/curve/encoding.rs:82-129):
// x from its 48 bytes, y = sqrt(x^3 + x + B) (Fp6::sqrt, src/fp6.rs:922-964), negated when its
// lexicographic rank disagrees with flag bit 6.  The subgroup test of AffinePoint::from_compressed
// (src/curve/affine.rs:303-309) runs afterwards as a fixed-scalar ladder that ANDs into is_some.
// A rejected record comes out as (0, 0, inf = 0) with is_some = 0.
__global__ void __launch_bounds__(BLOCK) k_decode_compressed(const u8* __restrict__ in, u64* __restrict__ xy, u8* __restrict__ inf,
                                                             u8* __restrict__ is_some, size_t n) {
    constexpr int REC = 49;
    __shared__ u32 s_buf[BLOCK * REC / 4];
    u8* sb = reinterpret_cast<u8*>(s_buf);
    const int t = threadIdx.x;
    const size_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t e0 = tile * BLOCK;
        const size_t cnt = n - e0 < (size_t)BLOCK ? n - e0 : (size_t)BLOCK;
        const size_t bytes = cnt * REC;
        const u8* src = in + e0 * REC;
        for (size_t w = t; w < bytes / 4; w += BLOCK) s_buf[w] = reinterpret_cast<const u32*>(src)[w];
        for (size_t b = (bytes / 4) * 4 + t; b < bytes; b += BLOCK) sb[b] = src[b];
        __syncthreads();
        if ((size_t)t < cnt) {
            const size_t e = e0 + t;
            const u8* rec = sb + t * REC;
            const bool inf_flag = (rec[48] >> 7) & 1, sort_flag = (rec[48] >> 6) & 1;
            Affine p = affine_identity();
            bool ok = lds_get_fp6(rec, p.x);
            if (ok && !(inf_flag && !sort_flag && fp6_is_zero(p.x))) {
                Fp6 y = fp6_sqrt(curve_rhs(p.x), &ok);
                if (fp6_lex_largest(y) != sort_flag) y = fp6_neg(y);
                p.y = y;
                p.inf = inf_flag;
                ok = ok && !inf_flag;
            }
            if (!ok) {
                p.x = fp6_zero();
                p.y = fp6_zero();
                p.inf = false;
            }
            store_fp6_vec_canonical(xy + 12 * e, p.x);
            store_fp6_vec_canonical(xy + 12 * e + 6, p.y);
            inf[e] = p.inf ? 1 : 0;
            is_some[e] = ok ? 1 : 0;
        }
        __syncthreads();
    }
}

// AffinePoint::from_x (src/curve/affine.rs:326-343): x = 0 is the identity, otherwise the point
// with the lexicographically largest y over x; none (0, 0, inf = 0, is_some = 0) when
// x^3 + x + B is not a square.
__global__ void __launch_bounds__(BLOCK) k_affine_from_x(const u64* __restrict__ x, u64* __restrict__ xy, u8* __restrict__ inf,
                                                         u8* __restrict__ is_some, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        Affine p;
        p.x = load_fp6_vec(x + 6 * i);
        bool ok = true;
        if (fp6_is_zero(p.x)) {
            p = affine_identity();
        } else {
            Fp6 y = fp6_sqrt(curve_rhs(p.x), &ok);
            if (!fp6_lex_largest(y)) y = fp6_neg(y);
            p.y = y;
            p.inf = false;
            if (!ok) {
                p.x = fp6_zero();
                p.y = fp6_zero();
            }
        }
        store_fp6_vec_canonical(xy + 12 * i, p.x);
        store_fp6_vec_canonical(xy + 12 * i + 6, p.y);
        inf[i] = p.inf ? 1 : 0;
        is_some[i] = ok ? 1 : 0;
    }
}

// The first half of {Affine,Projective,Jacobian}Point::random (src/curve/projective.rs:228-248, jacobian.rs:247-267,
// affine.rs:243-245) with the rng replaced by the seeded counter generator: element i owns the draws
// d(i, j) = splitmix64(seed, (start + i) * 256 + j); attempt a takes x = (d(i, 7a) % p, ..., d(i, 7a + 5) % p) (Fp6::random,
// src/fp6.rs:385-394) and flip_sign = d(i, 7a + 6) as a u32 is odd; the first attempt whose x^3 + x + B is a square gives
// (x, +-sqrt) -- the reference's loop.  cheetah_random_points_dev then clears the cofactor (fixed-scalar ladder).
// 36 attempts fit the 256 draws; an element that found none (probability 2^-36) leaves as the identity encoding (0, 1).
constexpr int RANDOM_MAX_ATTEMPTS = 36;
// Half of all x have no point over them, so a lane that retried on its own would hold its wave for the unluckiest of
// its 64 elements (7 - 8 attempts, measured: this kernel was 60 % of cheetah_random_points_dev).  Instead the wave
// shares the attempts out: every round its 64 lanes test attempts a .. a + per - 1 of each still unsolved element
// (per = 64 / unsolved: 1, then ~ 2, ~ 8, ...), the LOWEST successful attempt of an element wins -- the same point the
// sequential loop returns -- and a wave is done in 3 - 4 rounds.
__global__ void __launch_bounds__(BLOCK) k_random_candidates(u64 seed, u64 start, u64* __restrict__ xy, size_t n) {
    __shared__ u8 s_item[BLOCK];
    __shared__ int s_win[BLOCK];
    const int t = threadIdx.x, lane = t & 63, wbase = t & ~63;
    const size_t nround = (n + BLOCK - 1) / BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t i0 = round * BLOCK + wbase;  // the wave's elements: i0 .. i0 + 63
        u64 unsolved = __builtin_amdgcn_ballot_w64(i0 + lane < n);
        int next = 0;  // every unsolved element has tried attempts [0, next); wave-uniform
        while (unsolved != 0 && next < RANDOM_MAX_ATTEMPTS) {
            const int cnt = __builtin_popcountll(unsolved);
            int per = 64 / cnt;
            if (per > RANDOM_MAX_ATTEMPTS - next) per = RANDOM_MAX_ATTEMPTS - next;
            if ((unsolved >> lane) & 1ull) s_item[wbase + __builtin_popcountll(unsolved & ((1ull << lane) - 1ull))] = (u8)lane;
            s_win[t] = RANDOM_MAX_ATTEMPTS;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int slot = lane / per, a = next + lane % per;  // this lane: attempt a of the slot-th unsolved element
            const bool work = slot < cnt;
            const int item = work ? (int)s_item[wbase + slot] : 0;
            const u64 base = (start + i0 + (u64)item) * 256ull;
            Fp6 x;
#pragma unroll
            for (int l = 0; l < 6; l++) x.c[l] = fp_canonical(splitmix64(seed, base + 7 * a + l));
            const bool flip = (u32)splitmix64(seed, base + 7 * a + 6) & 1u;
            bool ok;
            Fp6 y = fp6_sqrt(curve_rhs(x), &ok);
            ok = ok && work;
            if (ok) atomicMin(&s_win[wbase + item], a);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (ok && s_win[wbase + item] == a) {  // the first success of its element, as the sequential loop would find it
                const size_t i = i0 + (size_t)item;
                store_fp6_vec_canonical(xy + 12 * i, x);
                store_fp6_vec_canonical(xy + 12 * i + 6, flip ? fp6_neg(y) : y);
            }
            unsolved &= ~__builtin_amdgcn_ballot_w64(s_win[t] != RANDOM_MAX_ATTEMPTS);
            next += per;
            __builtin_amdgcn_wave_barrier();
        }
        if ((unsolved >> lane) & 1ull) {  // no square in 36 attempts (2^-36): the identity encoding
            const size_t i = i0 + (size_t)lane;
            store_fp6_vec_canonical(xy + 12 * i, fp6_zero());
            store_fp6_vec_canonical(xy + 12 * i + 6, fp6_one());
        }
    }
}

// A record rejected by the subgroup check leaves as (0, 0, inf = 0) like every other none.
__global__ void __launch_bounds__(BLOCK) k_zero_rejected(u64* __restrict__ xy, u8* __restrict__ inf, const u8* __restrict__ is_some,
                                                         size_t n) {
    for (size_t w = (size_t)blockIdx.x * BLOCK + threadIdx.x; w < n * 12; w += (size_t)gridDim.x * BLOCK) {
        const size_t e = w / 12;
        if (!is_some[e]) {
            xy[w] = 0;
            if (w == e * 12) inf[e] = 0;
        }
    }
}

// is_on_curve: AffinePoint (FORM 0, src/curve/affine.rs:364-368: y^2 - x (x^2 + 1) == B, or the identity),
// ProjectivePoint (FORM 1, src/curve/projective.rs:940-946: Y^2 Z == X^3 + X Z^2 + B Z^3, or Z == 0),
// JacobianPoint (FORM 2, src/curve/jacobian.rs:930-938: Y^2 == X^3 + X Z^4 + B Z^6, or Z == 0).
template <int FORM>
__global__ void __launch_bounds__(BLOCK) k_is_on_curve(const u64* __restrict__ pts, u8* __restrict__ flags, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK)
        flags[i] = point_on_curve<FORM>(pts, i) ? 1 : 0;
}

// Complete group law over a batch: out[i] = lhs[i] + rhs[i] (LHS_FORM 1: ProjectivePoint::add / add_mixed,
// src/curve/projective.rs:456-569; 2: JacobianPoint::add / add_mixed, src/curve/jacobian.rs:452-552; RHS_FORM 0 = the
// mixed forms, rhs affine with (0, 1) as the identity's encoding), and out[i] = 2 pts[i] (::double, projective.rs:349-393,
// jacobian.rs:368-402).  The result is the reference's own representative, canonical limbs.
template <int LHS_FORM, int RHS_FORM>
__global__ void __launch_bounds__(BLOCK) k_add(const u64* __restrict__ lhs, const u64* __restrict__ rhs, u64* __restrict__ out,
                                               size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const Proj s = load_proj(lhs + 18 * i);
        Proj o;
        if (RHS_FORM == 0) {
            Affine r = load_affine(rhs + 12 * i);
            r.inf = affine_is_identity_encoding(r.x, r.y);
            o = LHS_FORM == 1 ? proj_add_mixed(s, r) : jac_add_mixed(s, r);
        } else {
            const Proj r = load_proj(rhs + 18 * i);
            o = LHS_FORM == 1 ? proj_add(s, r) : jac_add(s, r);
        }
        store_proj(out, nullptr, i, o);
    }
}
// ::double_multi(k): k >= 1 iterated ::double (src/curve/projective.rs:322-331, src/curve/jacobian.rs:341-350)
template <int FORM>
__global__ void __launch_bounds__(BLOCK) k_double(const u64* __restrict__ pts, u64* __restrict__ out, size_t n, u32 k) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        Proj s = load_proj(pts + 18 * i);
        for (u32 r = 0; r < k; r++) s = FORM == 1 ? proj_double(s) : jac_double(s);
        store_proj(out, nullptr, i, s);
    }
}
// The incomplete forms as operations of their own: ::double_multi_unchecked(k) (src/curve/projective.rs:337-346,
// src/curve/jacobian.rs:356-365) and ::add_mixed_unchecked (src/curve/projective.rs:576-610, src/curve/jacobian.rs:559-595).
template <int FORM>
__global__ void __launch_bounds__(BLOCK) k_double_unchecked(const u64* __restrict__ pts, u64* __restrict__ out, size_t n, u32 k) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        Proj s = load_proj(pts + 18 * i);
        for (u32 r = 0; r < k; r++) s = FORM == 1 ? proj_double_unchecked(s) : jac_double_unchecked(s);
        store_proj(out, nullptr, i, s);
    }
}
template <int FORM>
__global__ void __launch_bounds__(BLOCK) k_add_mixed_unchecked(const u64* __restrict__ lhs, const u64* __restrict__ rhs,
                                                               u64* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const Proj s = load_proj(lhs + 18 * i);
        Affine r = load_affine(rhs + 12 * i);
        r.inf = affine_is_identity_encoding(r.x, r.y);
        store_proj(out, nullptr, i, FORM == 1 ? proj_add_mixed_unchecked(s, r) : jac_add_mixed_unchecked(s, r));
    }
}
// -P: AffinePoint::neg (FORM 0, src/curve/affine.rs:164-175: the identity keeps y = 1), ProjectivePoint::neg
// (src/curve/projective.rs:447-453), JacobianPoint::neg (src/curve/jacobian.rs:443-449).  One Fp6 per thread: x and z
// pass through (canonicalised), y is negated.
template <int FORM>
__global__ void __launch_bounds__(BLOCK) k_neg(const u64* __restrict__ pts, u64* __restrict__ out, size_t n) {
    constexpr int W = FORM == 0 ? 2 : 3;  // Fp6 values per point
    for (size_t e = (size_t)blockIdx.x * BLOCK + threadIdx.x; e < n * W; e += (size_t)gridDim.x * BLOCK) {
        const size_t i = e / W;
        const int c = (int)(e - i * W);
        Fp6 v = load_fp6(pts + 6 * e);
        if (c == 1) {
            bool ident = false;
            if (FORM == 0) ident = affine_is_identity_encoding(load_fp6(pts + 6 * (e - 1)), v);
            v = fp6_select(fp6_neg(v), v, ident);
        }
        store_fp6_canonical(out + 6 * e, v);
    }
}
// flags[i] = a[i].ct_eq(&b[i]) for AffinePoint (src/curve/affine.rs:132-144), ProjectivePoint
// (src/curve/projective.rs:118-135), JacobianPoint (src/curve/jacobian.rs:134-154)
template <int FORM>
__global__ void __launch_bounds__(BLOCK) k_point_eq(const u64* __restrict__ a, const u64* __restrict__ b, u8* __restrict__ flags,
                                                    size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        bool eq;
        if (FORM == 0) {
            const Affine p = load_affine(a + 12 * i), q = load_affine(b + 12 * i);
            const bool pi = affine_is_identity_encoding(p.x, p.y), qi = affine_is_identity_encoding(q.x, q.y);
            eq = (pi && qi) || (!pi && !qi && fp6_eq(p.x, q.x) && fp6_eq(p.y, q.y));
        } else {
            const Proj p = load_proj(a + 18 * i), q = load_proj(b + 18 * i);
            eq = FORM == 1 ? proj_eq(p, q) : jac_eq(p, q);
        }
        flags[i] = eq ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
// Synthetic inputs: the generator of cheetah_amd/synth.py on the device.
__global__ void __launch_bounds__(BLOCK) k_synth_fp6(u64 seed, u64 start, u64* out, size_t n_words) {
    for (size_t w = (size_t)blockIdx.x * BLOCK + threadIdx.x; w < n_words; w += (size_t)gridDim.x * BLOCK)
        out[w] = fp_canonical(splitmix64(seed, 6 * start + w));
}
__global__ void __launch_bounds__(BLOCK) k_synth_scalars(u64 seed, u64 start, u64* out, size_t n) {
    // q = 0x7af2599b3b3f22d0563fbf0f990a37b5327aa72330157722d443623eaed4accf  (src/scalar.rs:41-46)
    const u64 q[4] = {0xd443623eaed4accfull, 0x327aa72330157722ull, 0x563fbf0f990a37b5ull, 0x7af2599b3b3f22d0ull};
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        u64 v[4];
        for (int j = 0; j < 4; j++) v[j] = splitmix64(seed, 4 * (start + i) + j);
        for (int rep = 0; rep < 3; rep++) {  // 2^256 / q < 3
            bool ge = true;  // v >= q ?
            for (int j = 3; j >= 0; j--) {
                if (v[j] != q[j]) {
                    ge = v[j] > q[j];
                    break;
                }
            }
            if (ge) {
                u64 borrow = 0;
                for (int j = 0; j < 4; j++) {
                    u64 d = v[j] - q[j] - borrow;
                    borrow = (v[j] < q[j] + borrow) || (q[j] + borrow < borrow) ? 1 : 0;
                    v[j] = d;
                }
            }
        }
        for (int j = 0; j < 4; j++) out[4 * i + j] = v[j];
    }
}

// ---------------------------------------------------------------------------
// Unit-level parity hook: one device function applied to n operand tuples.
__global__ void __launch_bounds__(BLOCK) k_debug_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        switch (op) {
            case 0: out[i] = fp_add(a[i], b[i]); break;
            case 1: out[i] = fp_sub(a[i], b[i]); break;
            case 2: out[i] = fp_mul(a[i], b[i]); break;
            case 3: out[i] = fp_mul_small(a[i], 7u); break;
            case 4: out[i] = fp_invert(a[i]); break;
            case 16: { Fp6 r = fp6_add(load_fp6(a + 6 * i), load_fp6(b + 6 * i)); for (int l = 0; l < 6; l++) out[6 * i + l] = r.c[l]; break; }
            case 17: { Fp6 r = fp6_sub(load_fp6(a + 6 * i), load_fp6(b + 6 * i)); for (int l = 0; l < 6; l++) out[6 * i + l] = r.c[l]; break; }
            case 18: { Fp6 r = fp6_mul_by_3b(load_fp6(a + 6 * i)); for (int l = 0; l < 6; l++) out[6 * i + l] = r.c[l]; break; }
            case 19: { Fp6 r = fp6_frobenius(load_fp6(a + 6 * i)); for (int l = 0; l < 6; l++) out[6 * i + l] = r.c[l]; break; }
            case 32: store_proj(out, nullptr, i, proj_double_unchecked(load_proj(a + 18 * i))); break;
            case 33: store_proj(out, nullptr, i, proj_add(load_proj(a + 18 * i), load_proj(b + 18 * i))); break;
            case 34: store_proj(out, nullptr, i, proj_add_mixed_unchecked(load_proj(a + 18 * i), load_affine(b + 12 * i))); break;
            case 35: {
                MJac p; p.x = load_fp6(a + 24 * i); p.y = load_fp6(a + 24 * i + 6); p.z = load_fp6(a + 24 * i + 12); p.z4 = load_fp6(a + 24 * i + 18);
                MJac r = mjac_double_unchecked(p);
                store_fp6_canonical(out + 24 * i, r.x); store_fp6_canonical(out + 24 * i + 6, r.y);
                store_fp6_canonical(out + 24 * i + 12, r.z); store_fp6_canonical(out + 24 * i + 18, r.z4);
                break;
            }
            case 36: {
                MJac p; p.x = load_fp6(a + 24 * i); p.y = load_fp6(a + 24 * i + 6); p.z = load_fp6(a + 24 * i + 12); p.z4 = load_fp6(a + 24 * i + 18);
                MJac r = mjac_add_mixed_unchecked(p, load_affine(b + 12 * i));
                store_fp6_canonical(out + 24 * i, r.x); store_fp6_canonical(out + 24 * i + 6, r.y);
                store_fp6_canonical(out + 24 * i + 12, r.z); store_fp6_canonical(out + 24 * i + 18, r.z4);
                break;
            }
            default: break;
        }
    }
}

// v_mad_u64_u32 issue-rate calibration (the roofline denominator): 8 independent
// chains per lane, 8 waves per SIMD.
__global__ void __launch_bounds__(BLOCK) k_mad_peak(u64* out, u32 seed, int iters) {
    u64 acc[8];
    u32 b = threadIdx.x * 2654435761u + seed, c = b ^ 0x9e3779b9u;
    for (int i = 0; i < 8; i++) acc[i] = ((u64)b << 32) + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(c) : "vcc");
        }
    }
    u64 r = 0;
    for (int i = 0; i < 8; i++) r ^= acc[i];
    if (r == 0x12345ull) out[0] = r;
}

}  // namespace cheetah
