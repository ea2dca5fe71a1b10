/*
 * cheetah_hip.h -- C ABI of the MI355X (gfx950) batched Cheetah curve engine.
 *
 * The reference (toposware/cheetah 0.1.0) is a leaf Rust crate with no FFI or
 * plugin surface (src/lib.rs:44 forbids unsafe); its API is one operation per
 * call on its own types.  This header is the batch boundary a thin Rust `-sys`
 * crate binds (INTEGRATION.md shows the binding): each entry point is the batch
 * form of the reference function cited beside it and returns, element by
 * element and in canonical limbs, what that function returns.
 *
 * Conventions
 *  - Fp6      : 6 x uint64_t, c0..c5, exactly Fp6::output_internal()        (src/fp6.rs:1126-1135)
 *  - affine   : 12 x uint64_t = x | y, plus one uint8_t infinity flag;
 *               the identity is x = 0, y = 1, infinity = 1                   (src/curve/affine.rs:224-230)
 *  - projective / jacobian : 18 x uint64_t = x | y | z                       (src/curve/projective.rs:213-225)
 *  - scalar   : 32 canonical little-endian bytes, Scalar::to_bytes()         (src/scalar.rs:477-489)
 *  - arrays are AoS, element i at index i; inputs may hold non-canonical
 *    limbs (any u64, as Fp::from_raw_unchecked, src/fp.rs:363-365); outputs are
 *    always canonical (each limb < p).
 *  - every call returns 0 on success or a negative cheetah_status; nothing
 *    aborts or throws across the boundary; cheetah_last_error() has the text.
 *  - a context belongs to ONE device and one host thread at a time; calls are
 *    issued on the context's stream.  `_dev` entry points take DEVICE pointers
 *    (8-byte aligned; Fp6 arrays 16-byte aligned) and are asynchronous on that stream; the others take HOST
 *    pointers, copy, run the same kernels and return after the results are back.
 *  - the caller owns all buffers; the library keeps no pointer after return.
 */
#ifndef CHEETAH_HIP_H
#define CHEETAH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cheetah_ctx cheetah_ctx;

typedef enum {
    CHEETAH_OK = 0,
    CHEETAH_ERR_INVALID = -1,   /* null pointer, bad enum value, misaligned device pointer */
    CHEETAH_ERR_HIP = -2,       /* a HIP runtime call failed (text in cheetah_last_error) */
    CHEETAH_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    CHEETAH_ERR_SCALAR = -4,    /* scalar outside what the reference's table covers (lookup.rs Q3) */
    CHEETAH_ERR_NOMEM = -5
} cheetah_status;

typedef enum { CHEETAH_AFFINE = 0, CHEETAH_PROJECTIVE = 1, CHEETAH_JACOBIAN = 2 } cheetah_form;

/* ---- context -------------------------------------------------------------
 * Creating a context builds on the device what the reference builds lazily on
 * first use: BASEPOINT_TABLE (src/constants.rs:21-22, src/lookup.rs:164-173),
 * BASEPOINT_LOOKUP (src/constants.rs:26-27), the shift point and the
 * MINUS_SHIFT_POINT_ARRAY entries the ladders use (src/constants.rs:43-90). */
int32_t cheetah_ctx_create(int device_id, cheetah_ctx **out);
int32_t cheetah_ctx_destroy(cheetah_ctx *ctx);
/* Issue all following work on an existing hipStream_t (e.g. torch's current stream).
 * NULL is HIP's default (null) stream, as everywhere in HIP.  A new context starts on a
 * private non-blocking stream; cheetah_ctx_use_own_stream() goes back to it. */
int32_t cheetah_ctx_set_stream(cheetah_ctx *ctx, void *hip_stream);
int32_t cheetah_ctx_use_own_stream(cheetah_ctx *ctx);
int32_t cheetah_ctx_synchronize(cheetah_ctx *ctx);
/* The ladder, fixed-base and normalize kernels are persistent: a launch fills every CU with as many workgroups as
 * fit and keeps them for the whole launch, so a kernel on another stream (an RCCL collective returning the previous
 * chunk's results) finds no free slot until a launch ends.  Reserving `cus` CUs' worth of slots (0 = none, the
 * default) sizes those launches smaller, so that such a kernel can always be placed.  No counterpart in the
 * reference (it has neither streams nor collectives); used by the multi-GPU path. */
int32_t cheetah_ctx_set_reserved_cus(cheetah_ctx *ctx, int cus);
const char *cheetah_last_error(const cheetah_ctx *ctx); /* ctx may be NULL: last create error */
const char *cheetah_version(void);

/* ---- elementwise Fp6 -----------------------------------------------------
 * out[i] = a[i] * b[i]           Fp6::mul      src/fp6.rs:651-834
 * out[i] = a[i]^2                Fp6::square   src/fp6.rs:838-920
 * out[i] = a[i]^-1, is_some[i]   Fp6::invert   src/fp6.rs:1035-1049 (0 -> out = 0, is_some = 0)
 * out[i] = sqrt(a[i]), is_some[i] Fp6::sqrt    src/fp6.rs:922-964 over Fp3::sqrt src/fp3.rs:247-277: the root the
 *                                reference's algorithm returns (not its negative); non-square -> out = 0, is_some = 0 */
int32_t cheetah_fp6_mul_batch(cheetah_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int32_t cheetah_fp6_square_batch(cheetah_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int32_t cheetah_fp6_invert_batch(cheetah_ctx *ctx, const uint64_t *a, uint64_t *out, uint8_t *is_some, size_t n);
int32_t cheetah_fp6_mul_batch_dev(cheetah_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int32_t cheetah_fp6_square_batch_dev(cheetah_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int32_t cheetah_fp6_invert_batch_dev(cheetah_ctx *ctx, const uint64_t *a, uint64_t *out, uint8_t *is_some, size_t n);
int32_t cheetah_fp6_sqrt_batch(cheetah_ctx *ctx, const uint64_t *a, uint64_t *out, uint8_t *is_some, size_t n);
int32_t cheetah_fp6_sqrt_batch_dev(cheetah_ctx *ctx, const uint64_t *a, uint64_t *out, uint8_t *is_some, size_t n);

/* ---- batch_normalize -----------------------------------------------------
 * xy[i], inf[i] = affine form of xyz[i]; identities (z = 0) give (0, 1, inf = 1).
 * ProjectivePoint::batch_normalize  src/curve/projective.rs:897-930
 * JacobianPoint::batch_normalize    src/curve/jacobian.rs:886-920
 * The reference's single sequential inversion chain is re-associated into
 * per-lane chunks; results are identical because field arithmetic is exact. */
int32_t cheetah_batch_normalize_projective(cheetah_ctx *ctx, const uint64_t *xyz, uint64_t *xy, uint8_t *inf, size_t n);
int32_t cheetah_batch_normalize_jacobian(cheetah_ctx *ctx, const uint64_t *xyz, uint64_t *xy, uint8_t *inf, size_t n);
int32_t cheetah_batch_normalize_projective_dev(cheetah_ctx *ctx, const uint64_t *xyz, uint64_t *xy, uint8_t *inf, size_t n);
int32_t cheetah_batch_normalize_jacobian_dev(cheetah_ctx *ctx, const uint64_t *xyz, uint64_t *xy, uint8_t *inf, size_t n);

/* ---- fixed base, constant time --------------------------------------------
 * out[i] = scalars[i] * G.
 * out_form = CHEETAH_AFFINE    : the affine point of &BASEPOINT_TABLE * &Scalar = BasePointTable::multiply
 *                                (src/lookup.rs:189-206, src/constants.rs:92-98); out = 12 limbs per
 *                                point, inf[i] set.  An affine point does not depend on the schedule
 *                                that reached it: this form runs a widened comb (signed radix-64,
 *                                43 additions, no doublings, 129 KiB table in LDS)
 * out_form = CHEETAH_PROJECTIVE: the reference's comb step for step; out = 18 limbs per point = the reference's own
 *                                (X:Y:Z) representative, canonical limbs; inf[i] = (Z == 0); inf may be NULL
 * out_form = CHEETAH_JACOBIAN  : &BASEPOINT_LOOKUP * &Scalar (src/constants.rs:102-135): radix-256
 *                                digits, LookupTable<128>, modified-Jacobian accumulator; out = 18
 *                                limbs = the reference's Jacobian (X, Y, Z); inf as for PROJECTIVE
 * BASEPOINT_TABLE: scalars whose top nibble exceeds 7 are outside LookupTable<8>'s range in the
 * reference: the call fails with CHEETAH_ERR_SCALAR (host entry point) or leaves
 * inf[i] = 0xFF for that element (device entry point).  BASEPOINT_LOOKUP takes any 32 bytes. */
int32_t cheetah_mul_by_generator_batch(cheetah_ctx *ctx, const uint8_t *scalars32, uint64_t *out, uint8_t *inf,
                                       size_t n, int out_form);
int32_t cheetah_mul_by_generator_batch_dev(cheetah_ctx *ctx, const uint8_t *scalars32, uint64_t *out, uint8_t *inf,
                                           size_t n, int out_form);

/* ---- variable base ---------------------------------------------------------
 * out[i] = scalars[i] * pts[i]
 * cheetah_multiply_batch          radix-16 signed windows, LookupTable<16>, constant-time table scan
 * cheetah_multiply_vartime_batch  width-5 NAF, NafLookupTable<8>, quirk Q1: always 256 doublings
 * The (in_form, out_form) pair names the reference method, as its own types do:
 *   PROJECTIVE on either side : ProjectivePoint::multiply / multiply_vartime  src/curve/projective.rs:614-662
 *   JACOBIAN on either side   : JacobianPoint::multiply / multiply_vartime    src/curve/jacobian.rs:599-649
 *   AFFINE -> AFFINE          : AffinePoint::multiply / multiply_vartime, which delegate to
 *                               JacobianPoint's (src/curve/affine.rs:373-386)
 *   PROJECTIVE with JACOBIAN  : no such method in the reference; CHEETAH_ERR_INVALID
 * in_form : CHEETAH_AFFINE (12 limbs; (0, 1) is read as the identity), CHEETAH_PROJECTIVE or
 *           CHEETAH_JACOBIAN (18 limbs).
 * out_form: AFFINE = 12 limbs + inf; PROJECTIVE / JACOBIAN = the reference's own (X, Y, Z) of that
 *           method, 18 canonical limbs, inf[i] = (Z == 0), inf may be NULL.
 * Any 32 bytes are accepted as a scalar, as in the reference. */
int32_t cheetah_multiply_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form, const uint8_t *scalars32,
                               uint64_t *out, uint8_t *inf, size_t n, int out_form);
int32_t cheetah_multiply_vartime_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form, const uint8_t *scalars32,
                                       uint64_t *out, uint8_t *inf, size_t n, int out_form);
int32_t cheetah_multiply_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int in_form, const uint8_t *scalars32,
                                   uint64_t *out, uint8_t *inf, size_t n, int out_form);
int32_t cheetah_multiply_vartime_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int in_form,
                                           const uint8_t *scalars32, uint64_t *out, uint8_t *inf, size_t n,
                                           int out_form);

/* cheetah_multiply_batch_dev for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same kernels
 * and bit-identical results; the per-lane window tables live in a file-scope device array of the library (96 MiB per
 * GPU, held until the process ends) instead of the context's hipMalloc'ed scratch, and the call makes no
 * hipMalloc / hipFree or any other allocating call.  The arena covers 128 workgroups, so large batches run at a
 * fraction of cheetah_multiply_batch_dev's rate; n > CHEETAH_NOALLOC_MAX_N is rejected with CHEETAH_ERR_INVALID.
 * All calls on one GPU share that GPU's arena (contents of the previous call are simply overwritten; nothing needs
 * clearing): two such calls that overlap in time on one GPU -- from different contexts or streams -- race. */
#define CHEETAH_NOALLOC_MAX_N (1u << 20)
int32_t cheetah_multiply_batch_dev_noalloc(cheetah_ctx *ctx, const uint64_t *pts, int in_form, const uint8_t *scalars32,
                                           uint64_t *out, uint8_t *inf, size_t n, int out_form);

/* ---- multi-scalar (Straus) ladders -----------------------------------------
 * out[g] = sum_{j < m} scalars[g*m + j] * pts[g*m + j]   (points and scalars interleaved per group)
 * cheetah_multiply_many_batch          {Projective,Jacobian}Point::multiply_many
 *                                      src/curve/projective.rs:817-832, src/curve/jacobian.rs:807-823;
 *                                      m = 2 is multiply_double (src/curve/projective.rs:666-685, jacobian.rs:653-673)
 * cheetah_multiply_many_vartime_batch  multiply_many_vartime (src/curve/projective.rs:840-869, jacobian.rs:831-858);
 *                                      m = 2 is multiply_double_vartime (:693-742 / :681-731, quirk Q1 included:
 *                                      both always run 256 doublings)
 * cheetah_multiply_double_with_basepoint_vartime_batch
 *                                      out[i] = s[2i] * pts[i] + s[2i+1] * G through ODD_MULTIPLES_BASEPOINT with a
 *                                      width-8 NAF (src/curve/projective.rs:756-809, src/curve/jacobian.rs:745-799)
 * Forms select Projective / Jacobian as for cheetah_multiply_batch (AffinePoint's wrappers give the same
 * affine point whichever ladder runs, src/curve/affine.rs:388-452).  1 <= m <= CHEETAH_MAX_MANY. */
#define CHEETAH_MAX_MANY 64
int32_t cheetah_multiply_many_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form, const uint8_t *scalars32, int m,
                                    uint64_t *out, uint8_t *inf, size_t n_groups, int out_form);
int32_t cheetah_multiply_many_vartime_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form, const uint8_t *scalars32,
                                            int m, uint64_t *out, uint8_t *inf, size_t n_groups, int out_form);
int32_t cheetah_multiply_double_with_basepoint_vartime_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form,
                                                             const uint8_t *scalars64, uint64_t *out, uint8_t *inf,
                                                             size_t n, int out_form);
int32_t cheetah_multiply_many_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int in_form, const uint8_t *scalars32, int m,
                                        uint64_t *out, uint8_t *inf, size_t n_groups, int out_form);
int32_t cheetah_multiply_many_vartime_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int in_form,
                                                const uint8_t *scalars32, int m, uint64_t *out, uint8_t *inf,
                                                size_t n_groups, int out_form);
int32_t cheetah_multiply_double_with_basepoint_vartime_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int in_form,
                                                                 const uint8_t *scalars64, uint64_t *out, uint8_t *inf,
                                                                 size_t n, int out_form);

/* ---- subgroup check, cofactor clearing ---------------------------------------
 * flags[i] = is_torsion_free(pts[i]) = (q * pts[i]).is_identity() through the vartime ladder
 *            (src/curve/projective.rs:885-893, src/curve/jacobian.rs:874-882, src/curve/affine.rs:463-467)
 * out[i]   = clear_cofactor(pts[i]) = multiply_vartime(pts[i], h)
 *            (src/curve/projective.rs:872-881, src/curve/jacobian.rs:861-870, src/curve/affine.rs:455-459) */
int32_t cheetah_is_torsion_free_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form, uint8_t *flags, size_t n);
int32_t cheetah_clear_cofactor_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form, uint64_t *out, uint8_t *inf,
                                     size_t n, int out_form);
int32_t cheetah_is_torsion_free_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int in_form, uint8_t *flags, size_t n);
int32_t cheetah_clear_cofactor_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int in_form, uint64_t *out, uint8_t *inf,
                                         size_t n, int out_form);

/* ---- point encodings (src/curve/encoding.rs) ------------------------------------
 * cheetah_encode_uncompressed_batch  AffinePoint::to_uncompressed = UncompressedPoint::from_affine   :169-183
 *                                    97 bytes: x | y canonical little-endian (zeros for the identity), flag byte bit 7 = infinity
 * cheetah_encode_compressed_batch    AffinePoint::to_compressed = CompressedPoint::from_affine       :55-77
 *                                    49 bytes: x, flag byte bit 7 = infinity, bit 6 = y.lexicographically_largest()
 *                                    (src/fp6.rs:398-424, evaluated exactly as written there)
 * cheetah_decode_uncompressed_batch  checked = 0: AffinePoint::from_uncompressed_unchecked = UncompressedPoint::to_affine  :187-226
 *                                    checked = 1: AffinePoint::from_uncompressed: also is_on_curve and is_torsion_free
 *                                                 (src/curve/affine.rs:289-292)
 *                                    is_some[i] = 0 marks a CtOption::none; its point is written as (0, 0, inf = 0).
 * cheetah_decode_compressed_batch    checked = 0: AffinePoint::from_compressed_unchecked = CompressedPoint::to_affine    :82-129
 *                                                 (y = Fp6::sqrt(x^3 + x + B), sign chosen by flag bit 6)
 *                                    checked = 1: AffinePoint::from_compressed: also is_torsion_free (src/curve/affine.rs:303-309)
 * cheetah_affine_from_x_batch        AffinePoint::from_x (src/curve/affine.rs:326-343): x = 0 -> identity, else the point with the
 *                                    lexicographically largest y; is_some = 0 when x^3 + x + B is not a square
 * `_dev`: xy 16-byte aligned, byte arrays 4-byte aligned. */
int32_t cheetah_encode_uncompressed_batch(cheetah_ctx *ctx, const uint64_t *xy, const uint8_t *inf, uint8_t *out97, size_t n);
int32_t cheetah_encode_compressed_batch(cheetah_ctx *ctx, const uint64_t *xy, const uint8_t *inf, uint8_t *out49, size_t n);
int32_t cheetah_decode_uncompressed_batch(cheetah_ctx *ctx, const uint8_t *in97, uint64_t *xy, uint8_t *inf, uint8_t *is_some,
                                          size_t n, int checked);
int32_t cheetah_decode_compressed_batch(cheetah_ctx *ctx, const uint8_t *in49, uint64_t *xy, uint8_t *inf, uint8_t *is_some,
                                        size_t n, int checked);
int32_t cheetah_affine_from_x_batch(cheetah_ctx *ctx, const uint64_t *x, uint64_t *xy, uint8_t *inf, uint8_t *is_some, size_t n);
int32_t cheetah_encode_uncompressed_batch_dev(cheetah_ctx *ctx, const uint64_t *xy, const uint8_t *inf, uint8_t *out97, size_t n);
int32_t cheetah_encode_compressed_batch_dev(cheetah_ctx *ctx, const uint64_t *xy, const uint8_t *inf, uint8_t *out49, size_t n);
int32_t cheetah_decode_uncompressed_batch_dev(cheetah_ctx *ctx, const uint8_t *in97, uint64_t *xy, uint8_t *inf,
                                              uint8_t *is_some, size_t n, int checked);
int32_t cheetah_decode_compressed_batch_dev(cheetah_ctx *ctx, const uint8_t *in49, uint64_t *xy, uint8_t *inf, uint8_t *is_some,
                                            size_t n, int checked);
int32_t cheetah_affine_from_x_batch_dev(cheetah_ctx *ctx, const uint64_t *x, uint64_t *xy, uint8_t *inf, uint8_t *is_some, size_t n);

/* ---- precomputed tables (read back for inspection / tests) ----------------
 * which: 0 = BASEPOINT_TABLE (256 affine points, table-major, src/lookup.rs:136),
 *        1 = shift point S, -2^4 S, -2^256 S (3 affine points, src/constants.rs:43-90),
 *        2 = BASEPOINT_LOOKUP = [G .. 128 G] (128 affine points, src/constants.rs:26-27),
 *        3 = ODD_MULTIPLES_BASEPOINT = [G, 3G .. 127 G] (64 affine points, src/constants.rs:32-33). */
int32_t cheetah_read_table(cheetah_ctx *ctx, int which, uint64_t *out_xy, size_t max_points);

/* ---- synthetic inputs generated on the device (bench / tests) -------------
 * The counter-based generator of cheetah_amd/synth.py: no reference counterpart
 * (the reference draws from OsRng, e.g. benches/fp6.rs:23-29). */
int32_t cheetah_synth_fp6_dev(cheetah_ctx *ctx, uint64_t seed, uint64_t start, uint64_t *out, size_t n);
int32_t cheetah_synth_scalars_dev(cheetah_ctx *ctx, uint64_t seed, uint64_t start, uint8_t *out32, size_t n);

/* ---- diagnostics -----------------------------------------------------------
 * Unit-level parity hook: applies ONE device function to n operand tuples.
 * No reference counterpart; used by tests/ to compare each device function
 * with the oracle.  op codes in cheetah_debug_op. */
typedef enum {
    CHEETAH_OP_FP_ADD = 0, CHEETAH_OP_FP_SUB, CHEETAH_OP_FP_MUL, CHEETAH_OP_FP_MUL7, CHEETAH_OP_FP_INVERT,
    CHEETAH_OP_FP6_ADD = 16, CHEETAH_OP_FP6_SUB, CHEETAH_OP_FP6_MUL3B, CHEETAH_OP_FP6_FROBENIUS,
    CHEETAH_OP_PROJ_DOUBLE = 32, CHEETAH_OP_PROJ_ADD, CHEETAH_OP_PROJ_ADD_MIXED_UNCHECKED,
    CHEETAH_OP_MJAC_DOUBLE, CHEETAH_OP_MJAC_ADD_MIXED_UNCHECKED
} cheetah_debug_op;
int32_t cheetah_debug_op_dev(cheetah_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* Measured v_mad_u64_u32 issue rate of this device in 1e12 lane-ops/s (roofline denominator). */
int32_t cheetah_calibrate_mad_peak(cheetah_ctx *ctx, double *tera_mads_per_s);
/* How many scalar multiplications one kernel launch of cheetah_multiply_batch_dev (vartime = 0) or
 * cheetah_multiply_vartime_batch_dev (vartime = 1) covers for a batch of n: the library bounds a launch to
 * a few rounds of the resident grid, so a large batch is a sequence of launches.  For per-launch rooflines
 * (bench.py); no counterpart in the reference. */
int32_t cheetah_multiply_units_per_launch(cheetah_ctx *ctx, size_t n, int vartime, int out_form, size_t *units);
/* Time of the last `_dev` kernel sequence bracketed by HIP events on the ctx stream. */
int32_t cheetah_timer_start(cheetah_ctx *ctx);
int32_t cheetah_timer_stop_ms(cheetah_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* CHEETAH_HIP_H */
