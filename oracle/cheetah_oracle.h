/*
 * cheetah_oracle.h -- CPU restatement of the toposware/cheetah hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the CPU baseline
 * ("kind": "port") of bench.py.  Nothing in the product path (cheetah_amd/,
 * include/) may include, link or call it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Every function follows the reference operation for operation and cites the
 * reference file:line it restates (paths relative to the reference crate).
 * Parity pinning: tests/test_oracle_kat.py checks this library against every
 * known-answer vector the reference's own tests hold for this path
 * (tests/golden/reference_kats.json) and against an independent big-integer
 * model (oracle/model.py).
 *
 * Layouts (all plain arrays, no padding surprises):
 *   Fp    : one uint64_t, any value is valid (non-canonical allowed)      src/fp.rs:57
 *   Fp6   : uint64_t[6], c0..c5 low degree first                          src/fp6.rs:49-62
 *   affine: uint64_t[12] = x[6] | y[6], plus a separate uint8_t infinity  src/curve/affine.rs:37-41
 *   proj / jacobian : uint64_t[18] = x | y | z                            src/curve/projective.rs:76-80
 *   modified jacobian: uint64_t[24] = x | y | z | z4                      src/curve/jacobian.rs:943-948
 *   scalar: uint8_t[32] canonical little-endian bytes                     src/scalar.rs:477-489
 */
#ifndef CHEETAH_ORACLE_H
#define CHEETAH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Fp (src/fp.rs) ---- */
uint64_t ora_fp_add(uint64_t a, uint64_t b);
uint64_t ora_fp_sub(uint64_t a, uint64_t b);
uint64_t ora_fp_neg(uint64_t a);
uint64_t ora_fp_double(uint64_t a);
uint64_t ora_fp_triple(uint64_t a);
uint64_t ora_fp_mul(uint64_t a, uint64_t b);
uint64_t ora_fp_square(uint64_t a);
uint64_t ora_fp_mul_by_u32(uint64_t a, uint32_t b);
uint64_t ora_fp_canonical(uint64_t a);
int ora_fp_invert(uint64_t a, uint64_t *out); /* returns is_some */
uint64_t ora_fp_exp(uint64_t a, uint64_t power);   /* src/fp.rs:304-333 */
int ora_fp_sqrt(uint64_t a, uint64_t *out);        /* src/fp.rs:199-229; returns is_some */
void ora_fp6_exp(const uint64_t *a, const uint64_t *power6, uint64_t *o); /* src/fp6.rs:1053-1086 */
void ora_fp6_exp_batch(const uint64_t *a, const uint64_t *power6, uint64_t *out, size_t n);
uint64_t ora_reduce_u128(uint64_t lo, uint64_t hi);
uint64_t ora_reduce_u96(uint64_t lo, uint32_t hi);

/* ---- Fp6 (src/fp6.rs) ---- */
void ora_fp6_add(const uint64_t *a, const uint64_t *b, uint64_t *out);
void ora_fp6_sub(const uint64_t *a, const uint64_t *b, uint64_t *out);
void ora_fp6_neg(const uint64_t *a, uint64_t *out);
void ora_fp6_double(const uint64_t *a, uint64_t *out);
void ora_fp6_triple(const uint64_t *a, uint64_t *out);
void ora_fp6_mul(const uint64_t *a, const uint64_t *b, uint64_t *out);
void ora_fp6_square(const uint64_t *a, uint64_t *out);
void ora_fp6_mul_by_u32(const uint64_t *a, uint32_t b, uint64_t *out);
void ora_fp6_mul_by_fp(const uint64_t *a, uint64_t b, uint64_t *out);
void ora_fp6_frobenius(const uint64_t *a, uint64_t *out);
void ora_fp6_frobenius_double(const uint64_t *a, uint64_t *out);
void ora_fp6_frobenius_triple(const uint64_t *a, uint64_t *out);
int ora_fp6_invert(const uint64_t *a, uint64_t *out); /* returns is_some; out = 0 if none */
void ora_fp6_canonical(const uint64_t *a, uint64_t *out);
int ora_fp6_is_zero(const uint64_t *a);
void ora_mul_by_3b(const uint64_t *a, uint64_t *out);

/* ---- scalar digit recoders (src/scalar.rs:531-643) ---- */
void ora_bytes_to_radix_16(const uint8_t *bytes32, int8_t *digits64);
void ora_bytes_to_radix_256(const uint8_t *bytes32, int8_t *digits32);
void ora_bytes_to_wnaf_vartime(const uint8_t *bytes32, int w, int8_t *naf256);

/* ---- point formulas ---- */
void ora_proj_double(const uint64_t *p, uint64_t *out);
void ora_proj_double_unchecked(const uint64_t *p, uint64_t *out);
void ora_proj_add(const uint64_t *p, const uint64_t *q, uint64_t *out);
void ora_proj_add_mixed(const uint64_t *p, const uint64_t *q_xy, uint8_t q_inf, uint64_t *out);
void ora_proj_add_mixed_unchecked(const uint64_t *p, const uint64_t *q_xy, uint8_t q_inf, uint64_t *out);
void ora_proj_to_affine(const uint64_t *p, uint64_t *out_xy, uint8_t *out_inf);
int ora_proj_is_on_curve(const uint64_t *p);

void ora_jac_double(const uint64_t *p, uint64_t *out);
void ora_jac_double_unchecked(const uint64_t *p, uint64_t *out);
void ora_jac_add_mixed_unchecked(const uint64_t *p, const uint64_t *q_xy, uint8_t q_inf, uint64_t *out);
void ora_jac_to_affine(const uint64_t *p, uint64_t *out_xy, uint8_t *out_inf);
int ora_jac_is_on_curve(const uint64_t *p);
void ora_mjac_double_unchecked(const uint64_t *p, uint64_t *out);
void ora_mjac_add_mixed_unchecked(const uint64_t *p, const uint64_t *q_xy, uint8_t q_inf, uint64_t *out);
void ora_mjac_to_affine(const uint64_t *p, uint64_t *out_xy, uint8_t *out_inf);

/* ---- batch_normalize (sequential chain exactly as the reference) ---- */
void ora_proj_batch_normalize(const uint64_t *p, size_t n, uint64_t *out_xy, uint8_t *out_inf);
void ora_jac_batch_normalize(const uint64_t *p, size_t n, uint64_t *out_xy, uint8_t *out_inf);

/* ---- tables ---- */
/* LookupTable<N>::from(&ProjectivePoint): out_xy = N*12 limbs, out_inf = N flags   src/lookup.rs:25-37 */
void ora_lookup_table_from_proj(const uint64_t *p, int n_entries, uint64_t *out_xy, uint8_t *out_inf);
/* NafLookupTable<N>::from(&ProjectivePoint)                                      src/naf_lookup.rs:35-48 */
void ora_naf_table_from_proj(const uint64_t *p, int n_entries, uint64_t *out_xy, uint8_t *out_inf);
/* global tables built on first use (lazy_static)                                 src/constants.rs:18-90 */
const uint64_t *ora_basepoint_table(void);       /* 32*8 affine points, 12 limbs each (no identities) */
const uint64_t *ora_basepoint_lookup(void);      /* LookupTable<128> of G, 128*12 limbs */
const uint64_t *ora_odd_multiples_basepoint(void); /* NafLookupTable<64> of G: G, 3G, ..., 127G; 64*12 limbs (src/constants.rs:32-33) */
const uint64_t *ora_minus_shift_point_array(void); /* 257 affine points, 12 limbs each */
const uint64_t *ora_shift_point_affine(void);    /* 12 limbs */
const uint64_t *ora_generator_affine(void);      /* 12 limbs */

/* ---- ladders: out is 18 limbs (projective / jacobian), raw (non-canonical) limbs ---- */
void ora_proj_multiply(const uint64_t *p, const uint8_t *by32, uint64_t *out);          /* projective.rs:614-625 */
void ora_proj_multiply_vartime(const uint64_t *p, const uint8_t *by32, uint64_t *out);  /* projective.rs:633-662 */
void ora_basepoint_table_multiply(const uint8_t *by32, uint64_t *out);                  /* lookup.rs:189-206 */
void ora_basepoint_lookup_multiply(const uint8_t *by32, uint64_t *out_jac);             /* constants.rs:124-135 */
void ora_jac_multiply(const uint64_t *p, const uint8_t *by32, uint64_t *out_jac);       /* jacobian.rs:599-611 */
void ora_jac_multiply_vartime(const uint64_t *p, const uint8_t *by32, uint64_t *out_jac); /* jacobian.rs:619-649 */

/* ---- batch drivers used by tests and by bench.py's cpu_baseline leg.
 * Independent elements are split statically over OpenMP threads (the analogue
 * of a rayon par_iter over the reference's scalar API).  Forms: 0 = affine
 * (12 limbs + inf flag), 1 = projective, 2 = Jacobian (18 limbs canonical).  A pair
 * with PROJECTIVE on either side runs ProjectivePoint::multiply*, any other pair
 * JacobianPoint::multiply* (what AffinePoint::multiply* delegates to);
 * ora_mul_by_generator_batch runs BASEPOINT_TABLE for forms 0 / 1 and
 * BASEPOINT_LOOKUP for form 2. ---- */
/* BasePointTable for a caller-supplied base: create (src/lookup.rs:164-173) and multiply (:189-206) */
void ora_basepoint_table_create(const uint64_t *p_in, int in_form, uint64_t *out_xy, uint8_t *out_inf); /* 256 x 12 limbs, 256 flags */
void ora_basepoint_table_multiply_batch(const uint64_t *table_xy, const uint8_t *table_inf, const uint8_t *scalars, uint64_t *out,
                                        uint8_t *out_inf, size_t n, int out_form);
int ora_affine_is_on_curve(const uint64_t *xy); /* src/curve/affine.rs:364-368 */
/* complete group law: JacobianPoint::add / add_mixed (src/curve/jacobian.rs:452-552), the two ct_eq (jacobian.rs:134-154,
 * projective.rs:118-135), and batch drivers over ProjectivePoint / JacobianPoint ::add, ::add_mixed, ::double */
void ora_jac_add(const uint64_t *p, const uint64_t *q, uint64_t *out);
void ora_jac_add_mixed(const uint64_t *p, const uint64_t *q_xy, uint8_t q_inf, uint64_t *out);
int ora_jac_eq(const uint64_t *p, const uint64_t *q);
int ora_proj_eq(const uint64_t *p, const uint64_t *q);
void ora_add_batch(const uint64_t *lhs, int lhs_form, const uint64_t *rhs, int rhs_form, uint64_t *out, size_t n);
void ora_double_batch(const uint64_t *pts, int form, uint64_t *out, size_t n);
void ora_double_multi_batch(const uint64_t *pts, int form, uint32_t k, uint64_t *out, size_t n); /* ::double_multi(k) */
void ora_double_multi_unchecked_batch(const uint64_t *pts, int form, uint32_t k, uint64_t *out, size_t n);
void ora_add_mixed_unchecked_batch(const uint64_t *lhs, int form, const uint64_t *rhs_xy, uint64_t *out, size_t n);
void ora_neg_batch(const uint64_t *pts, int form, uint64_t *out, size_t n);                      /* ::neg, three point types */
void ora_eq_batch(const uint64_t *a, const uint64_t *b, int form, uint8_t *flags, size_t n);    /* ::ct_eq, three point types */
int ora_num_threads(void);
void ora_fp6_mul_batch(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void ora_fp6_square_batch(const uint64_t *a, uint64_t *out, size_t n);
void ora_fp6_invert_batch(const uint64_t *a, uint64_t *out, uint8_t *is_some, size_t n);
void ora_mul_by_generator_batch(const uint8_t *scalars, uint64_t *out, uint8_t *out_inf, size_t n, int out_form);
void ora_multiply_batch(const uint64_t *pts, int in_form, const uint8_t *scalars, uint64_t *out,
                        uint8_t *out_inf, size_t n, int out_form);
void ora_multiply_vartime_batch(const uint64_t *pts, int in_form, const uint8_t *scalars, uint64_t *out,
                                uint8_t *out_inf, size_t n, int out_form);
/* Multi-scalar (Straus) ladders: out[g] = sum_j scalars[g*ns + j] * pts[g*m + j]  (+ scalars[g*ns + m] * G when
 * with_basepoint; ns = m + with_basepoint).  vartime = 0: multiply_double / multiply_many
 * (src/curve/projective.rs:666-685, :817-832, src/curve/jacobian.rs:653-673, :807-823); vartime = 1:
 * multiply_double_vartime / multiply_many_vartime / multiply_double_with_basepoint_vartime
 * (src/curve/projective.rs:693-809, :840-869, src/curve/jacobian.rs:681-858).  m <= 64. */
void ora_multiply_many_batch(const uint64_t *pts, int in_form, const uint8_t *scalars, int m, int vartime, int with_basepoint,
                             uint64_t *out, uint8_t *out_inf, size_t n_groups, int out_form);
/* ---- Fp3 and square roots (src/fp3.rs, src/fp6.rs:922-964) ---- */
int ora_fp3_mul(const uint64_t *a, const uint64_t *b, uint64_t *out);   /* src/fp3.rs:169-211; canonical out */
int ora_fp3_sqrt(const uint64_t *a, uint64_t *out);                     /* src/fp3.rs:255-282; returns is_some */
int ora_fp6_sqrt(const uint64_t *a, uint64_t *out);                     /* src/fp6.rs:922-964; returns is_some, out = 0 if none */
void ora_fp6_sqrt_batch(const uint64_t *a, uint64_t *out, uint8_t *is_some, size_t n);
int ora_compressed_to_affine(const uint8_t *in49, int checked, uint64_t *out_xy, uint8_t *out_inf); /* encoding.rs:82-129, affine.rs:300-309 */
void ora_decode_compressed_batch(const uint8_t *in, int checked, uint64_t *xy, uint8_t *inf, uint8_t *is_some, size_t n);
int ora_affine_from_x(const uint64_t *x6, uint64_t *out_xy, uint8_t *out_inf);                      /* src/curve/affine.rs:326-343 */
void ora_affine_from_x_batch(const uint64_t *x, uint64_t *xy, uint8_t *inf, uint8_t *is_some, size_t n);

/* ---- point encodings (src/curve/encoding.rs) ---- */
int ora_fp6_lexicographically_largest(const uint64_t *a);                                   /* src/fp6.rs:398-424 */
void ora_affine_to_compressed(const uint64_t *xy, uint8_t inf, uint8_t *out49);             /* encoding.rs:55-77 */
void ora_affine_to_uncompressed(const uint64_t *xy, uint8_t inf, uint8_t *out97);           /* encoding.rs:169-183 */
int ora_uncompressed_to_affine(const uint8_t *in97, int checked, uint64_t *out_xy, uint8_t *out_inf); /* encoding.rs:187-226, affine.rs:289-292 */
void ora_encode_batch(const uint64_t *xy, const uint8_t *inf, uint8_t *out, size_t n, int compressed);
void ora_decode_uncompressed_batch(const uint8_t *in, int checked, uint64_t *xy, uint8_t *inf, uint8_t *is_some, size_t n);
/* chunked: the reference call is one sequential chain per call; the batch driver
 * issues one reference-shaped call per `chunk` points per thread. */
void ora_jac_batch_normalize_chunked(const uint64_t *p, size_t n, size_t chunk, uint64_t *out_xy, uint8_t *out_inf);
void ora_proj_batch_normalize_chunked(const uint64_t *p, size_t n, size_t chunk, uint64_t *out_xy, uint8_t *out_inf);

#ifdef __cplusplus
}
#endif
#endif
