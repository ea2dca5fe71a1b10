// Goldilocks field Fp (p = 2^64 - 2^32 + 1) device arithmetic for gfx950.
//
// Semantics follow the reference's src/fp.rs: an element is ANY u64 (not kept
// canonical, src/fp.rs:57); results are congruent mod p to the reference's and
// are canonicalised (`% p`, src/fp.rs:125-127) only when they leave the GPU.
//
// Measured on MI355X (tools/valu_peak.hip, profiles/valu_peak_r01.json):
// v_mad_u64_u32 issues at the same rate as v_add_co_u32 / v_addc_co_u32
// (one wave64 instruction per 4 cycles per SIMD), so the cost of this path is
// the NUMBER of VALU instructions, not the number of multiplies.  The design
// therefore (a) multiplies schoolbook on 32-bit limbs, (b) accumulates many
// 32x32 products in 64-bit column accumulators with one carry add per product
// and (c) reduces mod p once per accumulated sum (Acc3 below) instead of once
// per product as the reference does (src/fp.rs:178-182, :623-646).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cheetah {

typedef uint32_t u32;
typedef uint64_t u64;

#define CH_DEV __device__ __forceinline__

constexpr u64 FP_P = 0xffffffff00000001ull;  // src/fp.rs:35
constexpr u64 FP_EPS = 0xffffffffull;        // src/fp.rs:42  (2^64 mod p)

CH_DEV u32 lo32(u64 x) { return (u32)x; }
CH_DEV u32 hi32(u64 x) { return (u32)(x >> 32); }
CH_DEV u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// a + b mod p; any u64 in, any u64 out.            (src/fp.rs:136-141)
CH_DEV u64 fp_add(u64 a, u64 b) {
    u64 s;
    bool c = __builtin_uaddll_overflow(a, b, (unsigned long long*)&s);
    u64 t;
    bool c2 = __builtin_uaddll_overflow(s, c ? FP_EPS : 0ull, (unsigned long long*)&t);
    return t + (c2 ? FP_EPS : 0ull);
}

// a - b mod p.                                     (src/fp.rs:163-168)
CH_DEV u64 fp_sub(u64 a, u64 b) {
    u64 s;
    bool c = __builtin_usubll_overflow(a, b, (unsigned long long*)&s);
    u64 t;
    bool c2 = __builtin_usubll_overflow(s, c ? FP_EPS : 0ull, (unsigned long long*)&t);
    return t - (c2 ? FP_EPS : 0ull);
}

CH_DEV u64 fp_neg(u64 a) { return fp_sub(0ull, a); }     // src/fp.rs:172-174
CH_DEV u64 fp_double(u64 a) { return fp_add(a, a); }     // src/fp.rs:145-150

// Three independent a + b / a - b at once, the form the Fp6 linear operations use.
// hipcc's code for fp_add / fp_sub re-derives each carry with a 64-bit compare (7 and 10
// VALU instructions); here the carry masks are used directly and eps = 2^32 - 1 is applied
// as carries (+ eps: low word - 1, high word + 1 less the low word's borrow): 6 VALU + 2 SALU
// per element.  The three chains are interleaved stage by stage so that two VALU
// instructions always separate the write of a carry mask from the carry-in that reads it
// (the two wait states gfx950 needs; the compiler pads its own code with s_nop).  The in/out operands are
// early-clobber: they are written while b operands of the other chains are still to be read, so no b operand
// may share a register with them (it could when both hold the same value, e.g. a shared constant 0).
#define CH_LIN3_OPERANDS                                                                                             \
    : [l0] "+&v"(l0), [h0] "+&v"(h0), [l1] "+&v"(l1), [h1] "+&v"(h1), [l2] "+&v"(l2), [h2] "+&v"(h2), [s0] "=&s"(s0), \
      [u0] "=&s"(u0), [s1] "=&s"(s1), [u1] "=&s"(u1), [s2] "=&s"(s2), [u2] "=&s"(u2)                                 \
    : [bl0] "v"(lo32(b0)), [bh0] "v"(hi32(b0)), [bl1] "v"(lo32(b1)), [bh1] "v"(hi32(b1)), [bl2] "v"(lo32(b2)),       \
      [bh2] "v"(hi32(b2))                                                                                            \
    : "scc"
CH_DEV void fp_add3(u64& a0, u64& a1, u64& a2, u64 b0, u64 b1, u64 b2) {
    u32 l0 = lo32(a0), h0 = hi32(a0), l1 = lo32(a1), h1 = hi32(a1), l2 = lo32(a2), h2 = hi32(a2);
    u64 s0, u0, s1, u1, s2, u2;
    asm("v_add_co_u32 %[l0], %[s0], %[l0], %[bl0]\n\t"
        "v_add_co_u32 %[l1], %[s1], %[l1], %[bl1]\n\t"
        "v_add_co_u32 %[l2], %[s2], %[l2], %[bl2]\n\t"
        "v_addc_co_u32 %[h0], %[s0], %[h0], %[bh0], %[s0]\n\t"
        "v_addc_co_u32 %[h1], %[s1], %[h1], %[bh1], %[s1]\n\t"
        "v_addc_co_u32 %[h2], %[s2], %[h2], %[bh2], %[s2]\n\t"
        // + c * eps: low word - c (borrow k in u), high word + (c & ~k) (carry c2 in u)
        "v_subbrev_co_u32 %[l0], %[u0], 0, %[l0], %[s0]\n\t"
        "v_subbrev_co_u32 %[l1], %[u1], 0, %[l1], %[s1]\n\t"
        "v_subbrev_co_u32 %[l2], %[u2], 0, %[l2], %[s2]\n\t"
        "s_andn2_b64 %[s0], %[s0], %[u0]\n\t"
        "s_andn2_b64 %[s1], %[s1], %[u1]\n\t"
        "s_andn2_b64 %[s2], %[s2], %[u2]\n\t"
        "v_addc_co_u32 %[h0], %[u0], %[h0], 0, %[s0]\n\t"
        "v_addc_co_u32 %[h1], %[u1], %[h1], 0, %[s1]\n\t"
        "v_addc_co_u32 %[h2], %[u2], %[h2], 0, %[s2]\n\t"
        // + c2 * eps (a third carry cannot occur)
        "v_subbrev_co_u32 %[l0], %[s0], 0, %[l0], %[u0]\n\t"
        "v_subbrev_co_u32 %[l1], %[s1], 0, %[l1], %[u1]\n\t"
        "v_subbrev_co_u32 %[l2], %[s2], 0, %[l2], %[u2]\n\t"
        "s_andn2_b64 %[u0], %[u0], %[s0]\n\t"
        "s_andn2_b64 %[u1], %[u1], %[s1]\n\t"
        "s_andn2_b64 %[u2], %[u2], %[s2]\n\t"
        "v_addc_co_u32 %[h0], %[s0], %[h0], 0, %[u0]\n\t"
        "v_addc_co_u32 %[h1], %[s1], %[h1], 0, %[u1]\n\t"
        "v_addc_co_u32 %[h2], %[s2], %[h2], 0, %[u2]" CH_LIN3_OPERANDS);
    a0 = mk64(l0, h0);
    a1 = mk64(l1, h1);
    a2 = mk64(l2, h2);
}
CH_DEV void fp_sub3(u64& a0, u64& a1, u64& a2, u64 b0, u64 b1, u64 b2) {
    u32 l0 = lo32(a0), h0 = hi32(a0), l1 = lo32(a1), h1 = hi32(a1), l2 = lo32(a2), h2 = hi32(a2);
    u64 s0, u0, s1, u1, s2, u2;
    asm("v_sub_co_u32 %[l0], %[s0], %[l0], %[bl0]\n\t"
        "v_sub_co_u32 %[l1], %[s1], %[l1], %[bl1]\n\t"
        "v_sub_co_u32 %[l2], %[s2], %[l2], %[bl2]\n\t"
        "v_subb_co_u32 %[h0], %[s0], %[h0], %[bh0], %[s0]\n\t"
        "v_subb_co_u32 %[h1], %[s1], %[h1], %[bh1], %[s1]\n\t"
        "v_subb_co_u32 %[h2], %[s2], %[h2], %[bh2], %[s2]\n\t"
        // - c * eps: low word + c (carry k in u), high word - (c & ~k) (borrow c2 in u)
        "v_addc_co_u32 %[l0], %[u0], %[l0], 0, %[s0]\n\t"
        "v_addc_co_u32 %[l1], %[u1], %[l1], 0, %[s1]\n\t"
        "v_addc_co_u32 %[l2], %[u2], %[l2], 0, %[s2]\n\t"
        "s_andn2_b64 %[s0], %[s0], %[u0]\n\t"
        "s_andn2_b64 %[s1], %[s1], %[u1]\n\t"
        "s_andn2_b64 %[s2], %[s2], %[u2]\n\t"
        "v_subbrev_co_u32 %[h0], %[u0], 0, %[h0], %[s0]\n\t"
        "v_subbrev_co_u32 %[h1], %[u1], 0, %[h1], %[s1]\n\t"
        "v_subbrev_co_u32 %[h2], %[u2], 0, %[h2], %[s2]\n\t"
        // - c2 * eps (a third borrow cannot occur)
        "v_addc_co_u32 %[l0], %[s0], %[l0], 0, %[u0]\n\t"
        "v_addc_co_u32 %[l1], %[s1], %[l1], 0, %[u1]\n\t"
        "v_addc_co_u32 %[l2], %[s2], %[l2], 0, %[u2]\n\t"
        "s_andn2_b64 %[u0], %[u0], %[s0]\n\t"
        "s_andn2_b64 %[u1], %[u1], %[s1]\n\t"
        "s_andn2_b64 %[u2], %[u2], %[s2]\n\t"
        "v_subbrev_co_u32 %[h0], %[s0], 0, %[h0], %[u0]\n\t"
        "v_subbrev_co_u32 %[h1], %[s1], 0, %[h1], %[u1]\n\t"
        "v_subbrev_co_u32 %[h2], %[s2], 0, %[h2], %[u2]" CH_LIN3_OPERANDS);
    a0 = mk64(l0, h0);
    a1 = mk64(l1, h1);
    a2 = mk64(l2, h2);
}
#undef CH_LIN3_OPERANDS

// a / 2 mod p: (a >> 1) + (a odd ? (p + 1) / 2 : 0); the sum stays below 2^64.
CH_DEV u64 fp_halve(u64 a) { return (a >> 1) + ((a & 1ull) ? 0x7fffffff80000001ull : 0ull); }

// Canonical representative in [0, p).               (src/fp.rs:125-127)
CH_DEV u64 fp_canonical(u64 a) { return a >= FP_P ? a - FP_P : a; }
CH_DEV bool fp_is_zero(u64 a) { return a == 0ull || a == FP_P; }
// 0 exactly when a is 0 or p = (1, 0xffffffff): the high word is then minus the low word and the low word is 0 or 1.
// Three full-rate instructions per limb and no compare, for OR-ing over several limbs (fp6_is_zero).
CH_DEV u32 fp_nonzero_bits(u64 a) { return (hi32(a) + lo32(a)) | (lo32(a) >> 1); }

// ---------------------------------------------------------------------------
// Column accumulators.  A sum of 64x64 products is kept as three 64-bit columns
// (weights 2^0, 2^32, 2^64), each with a 32-bit count of the carries out of it:
//   value = c0 + 2^32 c1 + 2^64 c2 + 2^64 t0 + 2^96 t1 + 2^128 t2
// One product costs 4 v_mad_u64_u32 + 4 v_addc_co_u32.  The carry of each mad
// is delivered in an SGPR pair, and gfx950 needs two wait states between a VALU
// write of an SGPR and a VALU read of it (the compiler pads its own code with
// s_nop for this; inside an asm block it is on us), so the eight instructions
// of a product are ordered such that two others always separate a mad from the
// addc that consumes its carry -- no s_nop is spent.
struct Acc3 {
    u64 c0, c1, c2;
    u32 t0, t1, t2;

    // acc = a * b
    CH_DEV void init(u64 a, u64 b) {
        u64 s0, s1, s2;
        asm("v_mad_u64_u32 %[c1], %[s1], %[al], %[bh], 0\n\t"
            "v_mad_u64_u32 %[c0], %[s0], %[al], %[bl], 0\n\t"
            "v_mad_u64_u32 %[c2], %[s2], %[ah], %[bh], 0\n\t"
            "v_mad_u64_u32 %[c1], %[s1], %[ah], %[bl], %[c1]\n\t"
            "v_mov_b32 %[t0], 0\n\t"
            "v_mov_b32 %[t2], 0\n\t"
            "v_addc_co_u32 %[t1], %[s1], 0, 0, %[s1]"
            : [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
              [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2)
            : [al] "v"(lo32(a)), [ah] "v"(hi32(a)), [bl] "v"(lo32(b)), [bh] "v"(hi32(b)));
    }
    // acc = a * b + c * d: init and the first mac in one block.  The carry counters are SET by the
    // first carries that can occur, so none needs zeroing first.
    CH_DEV void init2(u64 a, u64 b, u64 c, u64 d) {
        u64 s0, s1, s2;
        asm("v_mad_u64_u32 %[c1], %[s1], %[al], %[bh], 0\n\t"
            "v_mad_u64_u32 %[c0], %[s0], %[al], %[bl], 0\n\t"
            "v_mad_u64_u32 %[c1], %[s1], %[ah], %[bl], %[c1]\n\t"
            "v_mad_u64_u32 %[c2], %[s2], %[ah], %[bh], 0\n\t"
            "v_mad_u64_u32 %[c0], %[s0], %[cl], %[dl], %[c0]\n\t"
            "v_addc_co_u32 %[t1], %[s1], 0, 0, %[s1]\n\t"
            "v_mad_u64_u32 %[c1], %[s1], %[cl], %[dh], %[c1]\n\t"
            "v_mad_u64_u32 %[c2], %[s2], %[ch], %[dh], %[c2]\n\t"
            "v_addc_co_u32 %[t0], %[s0], 0, 0, %[s0]\n\t"
            "v_addc_co_u32 %[t1], %[s1], 0, %[t1], %[s1]\n\t"
            "v_mad_u64_u32 %[c1], %[s1], %[ch], %[dl], %[c1]\n\t"
            "v_addc_co_u32 %[t2], %[s2], 0, 0, %[s2]\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %[t1], %[s1], 0, %[t1], %[s1]"
            : [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
              [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2)
            : [al] "v"(lo32(a)), [ah] "v"(hi32(a)), [bl] "v"(lo32(b)), [bh] "v"(hi32(b)), [cl] "v"(lo32(c)),
              [ch] "v"(hi32(c)), [dl] "v"(lo32(d)), [dh] "v"(hi32(d)));
    }
    // acc += a * b
    CH_DEV void mac(u64 a, u64 b) {
        u64 s0, s1, s2;
        asm("v_mad_u64_u32 %[c1], %[s1], %[al], %[bh], %[c1]\n\t"
            "v_mad_u64_u32 %[c0], %[s0], %[al], %[bl], %[c0]\n\t"
            "v_mad_u64_u32 %[c2], %[s2], %[ah], %[bh], %[c2]\n\t"
            "v_addc_co_u32 %[t1], %[s1], 0, %[t1], %[s1]\n\t"
            "v_mad_u64_u32 %[c1], %[s1], %[ah], %[bl], %[c1]\n\t"
            "v_addc_co_u32 %[t0], %[s0], 0, %[t0], %[s0]\n\t"
            "v_addc_co_u32 %[t2], %[s2], 0, %[t2], %[s2]\n\t"
            "v_addc_co_u32 %[t1], %[s1], 0, %[t1], %[s1]"
            : [c0] "+&v"(c0), [c1] "+&v"(c1), [c2] "+&v"(c2), [t0] "+&v"(t0), [t1] "+&v"(t1), [t2] "+&v"(t2),
              [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2)
            : [al] "v"(lo32(a)), [ah] "v"(hi32(a)), [bl] "v"(lo32(b)), [bh] "v"(hi32(b)));
    }
    // acc = a * k0 + b * k1 for 32-bit k0, k1 (no column can overflow: t's are zero)
    CH_DEV void init_small2(u64 a, u32 k0, u64 b, u32 k1) {
        // each partial product < 2^64 - 2^33; columns 0 and 1 get two each
        u64 p0 = (u64)lo32(a) * k0, q0 = (u64)lo32(b) * k1;
        u64 p1 = (u64)hi32(a) * k0, q1 = (u64)hi32(b) * k1;
        bool cy0 = __builtin_uaddll_overflow(p0, q0, (unsigned long long*)&c0);
        bool cy1 = __builtin_uaddll_overflow(p1, q1, (unsigned long long*)&c1);
        t0 = cy0 ? 1u : 0u;
        t1 = cy1 ? 1u : 0u;
        c2 = 0;
        t2 = 0;
    }

    // The accumulated integer modulo p, as some u64.
    // With 2^64 = eps, 2^96 = -1, 2^128 = -2^32 (mod p) -- the fold of
    // src/fp.rs:623-646 applied once to the whole sum:
    //   w = 5-word integer (w4 small);  V = (w1:w0) + eps*w2 - (w3 + 2^32 w4)
    CH_DEV u64 reduce() const {
        // 5-word integer w4..w0 (w4 small) by two short carry chains
        u32 c;
        const u32 w0 = lo32(c0);
        const u32 w1 = __builtin_addc(hi32(c0), lo32(c1), 0u, &c);
        u32 w2 = __builtin_addc(hi32(c1), lo32(c2), c, &c);
        u32 w3 = __builtin_addc(hi32(c2), t1, c, &c);
        u32 w4 = t2 + c;
        w2 = __builtin_addc(w2, t0, 0u, &c);
        w3 = __builtin_addc(w3, 0u, c, &c);
        w4 += c;
        // R = (w1:w0) + eps * w2 in ONE v_mad_u64_u32 (eps = 0xffffffff is the inline constant -1);
        // its carry k (lane mask sk) is worth 2^64 = eps.
        u64 r = mk64(w0, w1), sk;
        asm("v_mad_u64_u32 %[r], %[sk], %[w2], -1, %[r]" : [r] "+v"(r), [sk] "=s"(sk) : [w2] "v"(w2));
        // Q = R - (w3 + 2^32 w4); its borrow b (mask sb) is worth -2^64 = -eps.  Net correction
        // (k - b) * eps; neither case can wrap.  + eps = + 2^32 - 1 and
        // - eps = - 2^32 + 1 are applied as carries: the low word gets -1 on the lanes of
        // sp = k & ~b and +1 on those of sn = b & ~k, the high word the opposite, each less the
        // carry / borrow that came out of the low word.  (SALU reads of a VALU-written mask are
        // interlocked by the hardware; a VALU carry-in read needs two wait states after the VALU
        // write, hence the one s_nop.)
        u32 q0, q1;
        u64 sb, sp, sn, sa;
        asm("v_sub_co_u32 %[q0], %[sb], %[r0], %[w3]\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32 %[q1], %[sb], %[r1], %[w4], %[sb]\n\t"
            "s_andn2_b64 %[sp], %[sk], %[sb]\n\t"
            "s_andn2_b64 %[sn], %[sb], %[sk]\n\t"
            "v_subbrev_co_u32 %[q0], %[sa], 0, %[q0], %[sp]\n\t"
            "v_addc_co_u32 %[q0], %[sb], %[q0], 0, %[sn]\n\t"
            "s_andn2_b64 %[sp], %[sp], %[sa]\n\t"
            "s_andn2_b64 %[sn], %[sn], %[sb]\n\t"
            "v_addc_co_u32 %[q1], %[sa], %[q1], 0, %[sp]\n\t"
            "v_subbrev_co_u32 %[q1], %[sa], 0, %[q1], %[sn]"
            : [q0] "=&v"(q0), [q1] "=&v"(q1), [sb] "=&s"(sb), [sp] "=&s"(sp), [sn] "=&s"(sn), [sa] "=&s"(sa)
            : [r0] "v"(lo32(r)), [r1] "v"(hi32(r)), [w3] "v"(w3), [w4] "v"(w4), [sk] "s"(sk)
            : "scc");
        return mk64(q0, q1);
    }
};

// a * b mod p.                                      (src/fp.rs:178-182)
CH_DEV u64 fp_mul(u64 a, u64 b) {
    Acc3 acc;
    acc.init(a, b);
    return acc.reduce();
}
CH_DEV u64 fp_square(u64 a) { return fp_mul(a, a); }  // src/fp.rs:194-196

// a * k mod p for a small constant k (k * 2^64 must fit the folds: k < 2^31).  (src/fp.rs:186-190)
CH_DEV u64 fp_mul_small(u64 a, u32 k) {
    u64 l = (u64)lo32(a) * k;             // < 2^63
    u64 h = (u64)hi32(a) * k + hi32(l);   // < 2^63; value = lo32(l) + 2^32 h
    // 2^32 * h = 2^32 lo32(h) + 2^64 hi32(h) = 2^32 lo32(h) + eps * hi32(h)
    u64 base = mk64(lo32(l), lo32(h));
    u64 e = (u64)hi32(h) * FP_EPS;        // hi32(h) < 2^31
    u64 r;
    bool c = __builtin_uaddll_overflow(base, e, (unsigned long long*)&r);
    return r + (c ? FP_EPS : 0ull);
}

CH_DEV u64 fp_triple(u64 a) { return fp_mul_small(a, 3u); }  // src/fp.rs:154-159

CH_DEV u64 fp_sqn(u64 a, int n) {
    for (int i = 0; i < n; i++) a = fp_square(a);
    return a;
}

// a^(p-2) by the reference's addition chain (src/fp.rs:335-359); 0 -> 0.
CH_DEV u64 fp_invert(u64 a) {
    u64 t2 = fp_square(a);
    u64 t3 = fp_mul(t2, a);
    u64 t0 = fp_sqn(t3, 2);
    t2 = fp_mul(t2, t0);
    t3 = fp_mul(t3, t0);
    t0 = fp_sqn(t3, 4);
    t2 = fp_mul(t2, t0);
    t3 = fp_mul(t3, t0);
    t0 = fp_sqn(t3, 8);
    t2 = fp_mul(t2, t0);
    t0 = fp_mul(t0, t3);
    t0 = fp_sqn(t0, 16);
    t2 = fp_mul(t2, t0);
    t0 = fp_sqn(t2, 32);
    t0 = fp_mul(t0, t2);
    return fp_mul(t0, a);
}

// a^power, the square-and-multiply of src/fp.rs:304-333 (exp selects, exp_vartime branches on the exponent bit: the
// same value; here the exponent is one value for the whole launch, so the branch is uniform).
CH_DEV u64 fp_exp(u64 a, u64 power) {
    u64 res = 1;
    for (int i = 63; i >= 0; i--) {
        res = fp_square(res);
        if ((power >> i) & 1) res = fp_mul(res, a);
    }
    return res;
}

// Fp::sqrt, src/fp.rs:199-229: Tonelli-Shanks for 2-adicity 32 in its constant-time form.  Returns the root the
// reference's algorithm returns (not its negative) and *is_some = (s^2 == a).
CH_DEV u64 fp_sqrt(u64 a, bool* is_some) {
    const u64 y = fp_exp(a, 0x7fffffffull);           // the progenitor a^((t - 1) / 2)
    u64 s = fp_mul(a, y), t = fp_mul(s, y);
    u64 z = 1753635133440165772ull;                   // TWO_ADIC_ROOT_OF_UNITY, src/fp.rs:48
    for (int k = 32; k >= 2; k--) {
        const u64 b = fp_sqn(t, k - 2);
        const bool is_one = fp_canonical(b) == 1;
        const u64 new_s = fp_mul(s, z);
        s = is_one ? s : new_s;
        z = fp_square(z);
        const u64 new_t = fp_mul(t, z);
        t = is_one ? t : new_t;
    }
    *is_some = fp_canonical(fp_square(s)) == fp_canonical(a);
    return s;
}

}  // namespace cheetah
