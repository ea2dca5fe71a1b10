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
 *  - a context belongs to one host thread at a time and to one device or several (cheetah_ctx_create);
 *    calls are issued on the context's stream.  `_dev` entry points take DEVICE pointers (memory of the
 *    context's first device; 8-byte aligned, Fp6 arrays 16-byte aligned) and are asynchronous on that stream;
 *    the others take HOST pointers and return after the results are back: per device a pipeline of chunks
 *    (upload / kernels / download overlapped; staging kept in the context and grown to the largest call seen, so a
 *    repeated call allocates nothing: two sets of device buffers and, for calls that move 8 MB or more per chunk, two
 *    sets of page-locked host buffers of at most 160 MB each, filled and emptied by up to four host threads of the
 *    library that live only for the duration of the call).
 *  - with several devices every call shards [0, n) into contiguous ranges, one per device, runs them
 *    concurrently and gathers the results into the caller's arrays (host arrays, or the first device's).
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
/* device_ids[0 .. n_dev): the HIP devices the context runs on (1 <= n_dev <= 64, no device twice, every id
 * below hipGetDeviceCount; anything else is CHEETAH_ERR_INVALID, no visible device CHEETAH_ERR_NO_DEVICE).  The
 * tables are replicated on every device.  n_dev = 1 is the plain single-GPU context. */
int32_t cheetah_ctx_create(const int *device_ids, int n_dev, cheetah_ctx **out);
int32_t cheetah_ctx_destroy(cheetah_ctx *ctx);
int32_t cheetah_ctx_num_devices(const cheetah_ctx *ctx);
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
/* A ladder or fixed-base call is issued as launches of at most `rounds` rounds of the resident grid (default 2: about
 * 2^19 scalar multiplications, 17 ms), so that a kernel on another stream waits for a launch boundary no longer than
 * that.  A caller with nothing else to place on the device may raise it (1 .. 256; 8 and more measure + 1 %).  No
 * counterpart in the reference. */
int32_t cheetah_ctx_set_launch_rounds(cheetah_ctx *ctx, int rounds);
const char *cheetah_last_error(const cheetah_ctx *ctx); /* ctx may be NULL: last create error */
const char *cheetah_version(void);

/* ---- the base field over a batch ----------------------------------------------------
 * out[i] = op(a[i] [, b[i] | scalar]) in Fp = GF(2^64 - 2^32 + 1), one u64 per element, any u64 accepted (src/fp.rs:57),
 * canonical output; b is read by ADD / SUB / MUL, scalar (low 32 bits) by MUL_BY_U32; INVERT maps 0 to 0 and, if is_some is
 * not NULL, sets is_some[i] = 0 there (CtOption::none) and 1 elsewhere:
 *   ADD SUB NEG DOUBLE TRIPLE   src/fp.rs:136-174      MUL SQUARE MUL_BY_U32   src/fp.rs:178-196 (reduce_u128 / reduce_u96, :623-667)
 *   INVERT                      src/fp.rs:335-359      CANONICAL (make_canonical)   src/fp.rs:125-127
 *   SQRT  Fp::sqrt (src/fp.rs:199-229): the root the reference's Tonelli-Shanks returns; a non-square gives out = 0 and
 *         is_some[i] = 0 (is_some may be NULL)         EXP  Fp::exp / exp_vartime (src/fp.rs:304-333): a[i]^scalar, scalar a u64 */
typedef enum {
    CHEETAH_FP_ADD = 0, CHEETAH_FP_SUB, CHEETAH_FP_NEG, CHEETAH_FP_DOUBLE, CHEETAH_FP_TRIPLE, CHEETAH_FP_MUL, CHEETAH_FP_SQUARE,
    CHEETAH_FP_MUL_BY_U32, CHEETAH_FP_INVERT, CHEETAH_FP_CANONICAL, CHEETAH_FP_SQRT, CHEETAH_FP_EXP
} cheetah_fp_op;
int32_t cheetah_fp_op_batch(cheetah_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, uint64_t scalar, uint64_t *out,
                            uint8_t *is_some, size_t n);
int32_t cheetah_fp_op_batch_dev(cheetah_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, uint64_t scalar, uint64_t *out,
                                uint8_t *is_some, size_t n);

/* ---- the other Fp6 operations over a batch ------------------------------------------
 * out[i] = op(a[i] [, b[i] | scalar]), canonical limbs; b is read by ADD / SUB only, scalar by MUL_BY_U32 (its low 32 bits)
 * and MUL_BY_FP (any u64, taken mod p):
 *   ADD SUB NEG DOUBLE TRIPLE            src/fp6.rs:968-1029          MUL_BY_U32 MUL_BY_FP   src/fp6.rs:428-464
 *   CANONICAL (make_canonical)           src/fp6.rs:373-382           MUL_BY_3B              src/curve/mod.rs:54-106
 *   FROBENIUS, FROBENIUS_DOUBLE, FROBENIUS_TRIPLE (x -> x^p, x^(p^2), x^(p^3))              src/fp6.rs:1088-1123 */
typedef enum {
    CHEETAH_FP6_ADD = 0, CHEETAH_FP6_SUB, CHEETAH_FP6_NEG, CHEETAH_FP6_DOUBLE, CHEETAH_FP6_TRIPLE, CHEETAH_FP6_MUL_BY_U32,
    CHEETAH_FP6_MUL_BY_FP, CHEETAH_FP6_CANONICAL, CHEETAH_FP6_FROBENIUS, CHEETAH_FP6_FROBENIUS_DOUBLE,
    CHEETAH_FP6_FROBENIUS_TRIPLE, CHEETAH_FP6_MUL_BY_3B
} cheetah_fp6_op;
int32_t cheetah_fp6_op_batch(cheetah_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, uint64_t scalar, uint64_t *out, size_t n);
int32_t cheetah_fp6_op_batch_dev(cheetah_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, uint64_t scalar, uint64_t *out, size_t n);

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
/* out[i] = a[i]^power       Fp6::exp / Fp6::exp_vartime   src/fp6.rs:1053-1086 (the two return the same element);
 * power6: HOST pointer (also for `_dev`) to the 6 little-endian 64-bit limbs of ONE exponent for the whole batch. */
int32_t cheetah_fp6_exp_batch(cheetah_ctx *ctx, const uint64_t *a, const uint64_t *power6, uint64_t *out, size_t n);
int32_t cheetah_fp6_exp_batch_dev(cheetah_ctx *ctx, const uint64_t *a, const uint64_t *power6, uint64_t *out, size_t n);

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
 * out[i] = scalars[i] * G.  Constant time on the device means: no branch and no memory address depends on a scalar;
 * a table lookup goes over every entry of the window's table at addresses all lanes share (the reference's scan,
 * src/lookup.rs:81-102), and only which lanes are enabled for which entry's read depends on the digit.
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
 * BASEPOINT_TABLE (AFFINE and PROJECTIVE alike): scalars whose top nibble exceeds 7 are outside LookupTable<8>'s
 * range in the reference (a debug_assert there).  The host entry point fails with CHEETAH_ERR_SCALAR
 * before anything runs; the device entry point, which cannot look at the scalars, rejects the ELEMENT: its limbs are
 * written as zeros and inf[i] = 0xFF, the same for both forms -- so inf must not be NULL for these two forms.
 * BASEPOINT_LOOKUP (JACOBIAN) takes any 32 bytes, and its inf may be NULL. */
int32_t cheetah_mul_by_generator_batch(cheetah_ctx *ctx, const uint8_t *scalars32, uint64_t *out, uint8_t *inf,
                                       size_t n, int out_form);
int32_t cheetah_mul_by_generator_batch_dev(cheetah_ctx *ctx, const uint8_t *scalars32, uint64_t *out, uint8_t *inf,
                                           size_t n, int out_form);

/* ---- fixed base, caller-supplied point ---------------------------------------
 * cheetah_basepoint_table_create   BasePointTable::create(&P) / BasePointTable::from(P)    src/lookup.rs:138-173
 *                                  point: HOST pointer to ONE point, in_form AFFINE (12 limbs; (0, 1) is the identity)
 *                                  or PROJECTIVE (18 limbs).  Builds on every device of the context the 32 x
 *                                  LookupTable<8> of 256^i P (24 KiB) and the widened table of the AFFINE form.
 * cheetah_basepoint_table_multiply_batch[_dev]
 *                                  out[i] = scalars[i] * P = BasePointTable::multiply        src/lookup.rs:189-206
 *                                  (multiply_vartime, :216-233, is the same body -- SURVEY Q2 -- hence the same call).
 *                                  out_form PROJECTIVE: the reference's comb step for step, its own (X:Y:Z);
 *                                  AFFINE: the affine point by the widened comb (a base that is NOT on the curve,
 *                                  which create() detects, takes the reference's comb and a normalisation instead,
 *                                  so that the value stays the reference's).  Range rule and inf as for
 *                                  cheetah_mul_by_generator_batch.  With P = G this is `&BASEPOINT_TABLE * &Scalar`
 *                                  (the relation the reference tests, src/curve/projective.rs:1328-1351).
 * cheetah_basepoint_table_read     the 256 affine entries, table-major (tests; get_basepoint() is entry 0, :176-178)
 * Like the reference's ladders the comb adds with the incomplete formulas from the shift point: a base chosen
 * against the shift point (P = +-2^j S ...) is outside what either implementation promises. */
typedef struct cheetah_basepoint_table cheetah_basepoint_table;
int32_t cheetah_basepoint_table_create(cheetah_ctx *ctx, const uint64_t *point, int in_form, cheetah_basepoint_table **out);
int32_t cheetah_basepoint_table_destroy(cheetah_basepoint_table *table);
int32_t cheetah_basepoint_table_multiply_batch(cheetah_ctx *ctx, const cheetah_basepoint_table *table, const uint8_t *scalars32,
                                               uint64_t *out, uint8_t *inf, size_t n, int out_form);
int32_t cheetah_basepoint_table_multiply_batch_dev(cheetah_ctx *ctx, const cheetah_basepoint_table *table,
                                                   const uint8_t *scalars32, uint64_t *out, uint8_t *inf, size_t n, int out_form);
int32_t cheetah_basepoint_table_read(cheetah_ctx *ctx, const cheetah_basepoint_table *table, uint64_t *out_xy, size_t max_points);

/* ---- variable base ---------------------------------------------------------
 * out[i] = scalars[i] * pts[i]
 * cheetah_multiply_batch          radix-16 signed windows, LookupTable<16>, constant-time table scan
 * cheetah_multiply_vartime_batch  width-5 NAF, NafLookupTable<8>, quirk Q1: always 256 doublings
 * The (in_form, out_form) pair names the reference method whose RESULT is returned, as its own types do:
 *   out_form PROJECTIVE       : ProjectivePoint::multiply / multiply_vartime  src/curve/projective.rs:614-662,
 *                               its own (X:Y:Z): the projective accumulator and, for vartime, its NAF sequence per lane
 *   out_form JACOBIAN         : JacobianPoint::multiply / multiply_vartime    src/curve/jacobian.rs:599-649,
 *                               its own (X, Y, Z): the modified-Jacobian accumulator, same schedules
 *   out_form AFFINE           : the affine point of the method the in_form names (AffinePoint::multiply* delegate to
 *                               JacobianPoint's, src/curve/affine.rs:373-386).  An affine point does not depend on the
 *                               schedule that reached it, so the cheapest one runs: the modified-Jacobian accumulator
 *                               also for PROJECTIVE inputs, and for multiply_vartime the fixed-window ladder with
 *                               indexed (variable-time) table reads instead of the NAF ladder -- except for scalars
 *                               with bit 255 set, where the two differ and the NAF ladder runs
 *   PROJECTIVE with JACOBIAN  : no such method in the reference; CHEETAH_ERR_INVALID
 * in_form : CHEETAH_AFFINE (12 limbs; (0, 1) is read as the identity), CHEETAH_PROJECTIVE or
 *           CHEETAH_JACOBIAN (18 limbs).
 * out_form: AFFINE = 12 limbs + inf; PROJECTIVE / JACOBIAN = the reference's own (X, Y, Z) of that
 *           method, 18 canonical limbs, inf[i] = (Z == 0), inf may be NULL.
 * Any 32 bytes are accepted as a scalar, as in the reference.
 * Constant time, for cheetah_multiply_batch, means: no branch and no memory address in the window loops depends on a
 * digit of a canonical scalar (re-checked on the ISA of every build, tools/check_ct_isa.py).  It does NOT mean
 * independence of the POINTS: a wave that holds the identity, a point of order <= 16 (or one with the point of order 2
 * among its first 16 multiples: orders 2 and 10) or a point that is not on the
 * curve rebuilds its window tables with the reference's complete additions (src/lookup.rs:25-37) instead of the
 * shared-inversion affine steps, and a wave that holds a scalar with the top nibble > 7 (no canonical scalar) builds
 * 16 table entries instead of 8.  The reference's LookupTable::from has neither branch;
 * cheetah_ctx_set_point_independent_tables removes the first.
 * cheetah_multiply_vartime_batch with a PROJECTIVE / JACOBIAN out_form runs a scheduling loop (every wave regroups its
 * 128 items by the step they want next); the loop has a hard bound on its phases, far above what any scalar needs
 * (tools/naf_pool_sim.py, tests/test_naf_pool_cpu.py), so that a defect could never spin on the GPU: an item left
 * unfinished at the bound would be returned as zero limbs with inf[i] = 0xFE.  No input produces that value.
 * Points are not required to be on the curve (the reference does not check either): such inputs return what the
 * reference's formulas return for them, by the reference's own table construction and, for the vartime entry
 * points, its own NAF schedule (tests/test_gpu_boundary.py). */
int32_t cheetah_multiply_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form, const uint8_t *scalars32,
                               uint64_t *out, uint8_t *inf, size_t n, int out_form);
/* on != 0: the constant-time ladders (cheetah_multiply_batch, cheetah_multiply_many_batch and their _dev / _noalloc
 * forms) build every wave's window tables by the reference's own construction (LookupTable::from, complete additions,
 * src/lookup.rs:25-37), so that no wave-level choice depends on the points either; same results, 4 % slower
 * (measured at 2^22: 29.2 against 30.3 M/s, profiles/r03_point_independent_tables.txt).
 * Off by default.  Applies to every device of the context; blocks until the setting is in place. */
int32_t cheetah_ctx_set_point_independent_tables(cheetah_ctx *ctx, int on);
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
 * affine point whichever ladder runs, src/curve/affine.rs:388-452).  1 <= m <= CHEETAH_MAX_MANY.
 * With an AFFINE out_form (level L1: the affine point, which no schedule can change) the two vartime entry points do
 * not run the NAF schedule: multiply_many_vartime runs the fixed-window Straus ladder with table entries read by
 * index, and multiply_double_with_basepoint_vartime is a * P by the fixed-window ladder plus b * G by the widened comb
 * plus one complete addition.  Scalars with bit 255 set (no canonical scalar has it; the reference's NAF recoding
 * truncates there) are detected per wave of 64 consecutive elements and those waves take the reference's schedule. */
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

/* ---- curve membership ------------------------------------------------------------
 * flags[i] = is_on_curve(pts[i]):  AffinePoint     src/curve/affine.rs:364-368  (y^2 - x (x^2 + 1) == B, or the identity (0, 1))
 *                                  ProjectivePoint src/curve/projective.rs:940-946 (Y^2 Z == X^3 + X Z^2 + B Z^3, or Z == 0)
 *                                  JacobianPoint   src/curve/jacobian.rs:930-938   (Y^2 == X^3 + X Z^4 + B Z^6, or Z == 0) */
int32_t cheetah_is_on_curve_batch(cheetah_ctx *ctx, const uint64_t *pts, int in_form, uint8_t *flags, size_t n);
int32_t cheetah_is_on_curve_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int in_form, uint8_t *flags, size_t n);

/* ---- group law on batches (the complete formulas) -----------------------------------
 * out[i] = lhs[i] + rhs[i], lhs and out as n x 18 limbs in lhs_form, which is CHEETAH_PROJECTIVE or CHEETAH_JACOBIAN:
 *   rhs_form == lhs_form      ProjectivePoint::add        src/curve/projective.rs:456-515
 *                             JacobianPoint::add          src/curve/jacobian.rs:452-505
 *   rhs_form == CHEETAH_AFFINE  ProjectivePoint::add_mixed  src/curve/projective.rs:518-569
 *                             JacobianPoint::add_mixed    src/curve/jacobian.rs:508-552
 *                             (rhs as n x 12 limbs, (0, 1) = the identity, as for every affine input here)
 * out[i] = 2 pts[i]           ProjectivePoint::double     src/curve/projective.rs:349-393
 *                             JacobianPoint::double       src/curve/jacobian.rs:368-402
 * Parity level L2: out[i] is the representative (X : Y : Z) the reference's method returns, canonical limbs,
 * including which representative its special cases select (JacobianPoint::identity() = (1, 1, 0) for P + (-P),
 * self.double() for P + P, the other operand as is when one is the identity).  Inputs need not be canonical and are
 * not checked for curve membership (the reference's methods do not check either).  out may alias lhs or pts. */
int32_t cheetah_add_batch(cheetah_ctx *ctx, const uint64_t *lhs, int lhs_form, const uint64_t *rhs, int rhs_form,
                          uint64_t *out, size_t n);
int32_t cheetah_add_batch_dev(cheetah_ctx *ctx, const uint64_t *lhs, int lhs_form, const uint64_t *rhs, int rhs_form,
                              uint64_t *out, size_t n);
int32_t cheetah_double_batch(cheetah_ctx *ctx, const uint64_t *pts, int form, uint64_t *out, size_t n);
int32_t cheetah_double_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int form, uint64_t *out, size_t n);
/* out[i] = 2^k pts[i] by k iterated ::double, k >= 1 (the reference asserts it):
 *                             ProjectivePoint::double_multi  src/curve/projective.rs:322-331
 *                             JacobianPoint::double_multi    src/curve/jacobian.rs:341-350 */
int32_t cheetah_double_multi_batch(cheetah_ctx *ctx, const uint64_t *pts, int form, uint32_t k, uint64_t *out, size_t n);
int32_t cheetah_double_multi_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int form, uint32_t k, uint64_t *out, size_t n);
/* The incomplete ("unchecked") formulas as operations of their own, for callers that chain them as the reference's
 * ladders do; form CHEETAH_PROJECTIVE or CHEETAH_JACOBIAN, L2 results.  As in the reference they are only meaningful
 * away from their exceptional cases (the identity for the doublings; equal, opposite or identity operands on the
 * left for the addition -- an identity rhs (0, 1) returns lhs):
 *   out[i] = 2^k pts[i], k >= 1   ::double_multi_unchecked / ::double_unchecked (k = 1)
 *                                 src/curve/projective.rs:337-346, :400-443; src/curve/jacobian.rs:356-365, :409-439
 *   out[i] = lhs[i] + rhs[i]      ::add_mixed_unchecked, rhs affine (n x 12)
 *                                 src/curve/projective.rs:576-610; src/curve/jacobian.rs:559-595 */
int32_t cheetah_double_multi_unchecked_batch(cheetah_ctx *ctx, const uint64_t *pts, int form, uint32_t k, uint64_t *out, size_t n);
int32_t cheetah_double_multi_unchecked_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int form, uint32_t k, uint64_t *out, size_t n);
int32_t cheetah_add_mixed_unchecked_batch(cheetah_ctx *ctx, const uint64_t *lhs, int form, const uint64_t *rhs_xy, uint64_t *out, size_t n);
int32_t cheetah_add_mixed_unchecked_batch_dev(cheetah_ctx *ctx, const uint64_t *lhs, int form, const uint64_t *rhs_xy, uint64_t *out, size_t n);
/* out[i] = -pts[i], form AFFINE (n x 12), PROJECTIVE or JACOBIAN (n x 18), canonical limbs:
 *                             AffinePoint::neg      src/curve/affine.rs:164-175  (the identity (0, 1) stays (0, 1))
 *                             ProjectivePoint::neg  src/curve/projective.rs:447-453
 *                             JacobianPoint::neg    src/curve/jacobian.rs:443-449 */
int32_t cheetah_neg_batch(cheetah_ctx *ctx, const uint64_t *pts, int form, uint64_t *out, size_t n);
int32_t cheetah_neg_batch_dev(cheetah_ctx *ctx, const uint64_t *pts, int form, uint64_t *out, size_t n);
/* flags[i] = 1 if a[i] and b[i] are the same point (whatever their representatives), else 0; both arrays in `form`:
 *                             ConstantTimeEq for AffinePoint      src/curve/affine.rs:132-144
 *                                                ProjectivePoint  src/curve/projective.rs:118-135
 *                                                JacobianPoint    src/curve/jacobian.rs:134-154 */
int32_t cheetah_eq_batch(cheetah_ctx *ctx, const uint64_t *a, const uint64_t *b, int form, uint8_t *flags, size_t n);
int32_t cheetah_eq_batch_dev(cheetah_ctx *ctx, const uint64_t *a, const uint64_t *b, int form, uint8_t *flags, size_t n);

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
/* Seeded points of the prime-order subgroup shaped like the reference's `random()`:
 *   ProjectivePoint::random  src/curve/projective.rs:228-248   JacobianPoint::random  src/curve/jacobian.rs:247-267
 *   AffinePoint::random      src/curve/affine.rs:243-245       (out_form selects; PROJECTIVE / JACOBIAN give the
 *   representative clear_cofactor() = multiply_vartime(h) returns, AFFINE its affine point)
 * with the rng replaced by the counter generator: element i draws d(i, j) = splitmix64(seed, (start + i) * 256 + j); attempt
 * a = 0, 1, ... takes x = (d(i, 7a) % p, ..., d(i, 7a + 5) % p) (Fp6::random) and flip_sign = low 32 bits of d(i, 7a + 6) odd;
 * the first attempt whose x^3 + x + B is a square gives (x, y or -y), y = Fp6::sqrt(...); then clear_cofactor.  The
 * reference retries on the identity; here that element (probability ~ 2^-255), or one with no square among its 36
 * attempts (2^-36), is returned as the identity (inf = 1).  _dev: device pointers, 16-byte aligned; the other form
 * fills HOST arrays with the same stream (blocking).  inf is required for AFFINE and may be NULL otherwise. */
int32_t cheetah_random_points_dev(cheetah_ctx *ctx, uint64_t seed, uint64_t start, uint64_t *out, uint8_t *inf, size_t n, int out_form);
int32_t cheetah_random_points(cheetah_ctx *ctx, uint64_t seed, uint64_t start, uint64_t *out, uint8_t *inf, size_t n, int out_form);

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
/* A multi-shard context whose n_shards shards all live on ONE device: the sharding, peer-copy and gather paths of a
 * multi-device context on a one-GPU box (tests only; cheetah_ctx_create itself rejects a device listed twice). */
int32_t cheetah_debug_ctx_create_replicated(int device_id, int n_shards, cheetah_ctx **out);
/* on != 0: multiply_double_with_basepoint_vartime with AFFINE results runs the reference's Straus NAF schedule
 * (k_multiply_many_vartime) instead of the composed one (a P by the fixed-window ladder + b G by the widened comb + one
 * complete addition).  The affine results are the same; tests and tools compare and time the two. */
int32_t cheetah_debug_set_reference_schedule(cheetah_ctx *ctx, int on);
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
