// C ABI of the batched Cheetah engine (include/cheetah_hip.h).  Host side only:
// argument checks, device buffers, kernel launches.  No CPU fallback exists: a
// call either runs the HIP kernels or returns an error code.
#include "../../include/cheetah_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "kernels.hip.h"

using namespace cheetah;

// ---- the allocation-free entry point's arena (cheetah_multiply_batch_dev_noalloc) ----------------------
// A file-scope device array instead of the context's hipMalloc'ed scratch: window tables for
// NOALLOC_MAX_BLOCKS workgroups of k_multiply_ct (96 MiB; every GPU the library runs on holds its own
// copy until the process ends).  In host code the array's NAME is only a handle: its device address is
// taken once per device with hipGetSymbolAddress at context creation (cheetah_ctx::noalloc_arena) and
// only that pointer is ever passed to a kernel.
constexpr int NOALLOC_MAX_BLOCKS = 128;
constexpr size_t NOALLOC_WORDS = (size_t)NOALLOC_MAX_BLOCKS * LaneScratch::WORDS_PER_BLOCK;
static __device__ __align__(256) u64 noalloc_scratch[NOALLOC_WORDS];

struct cheetah_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    DeviceTables* tables = nullptr;
    u8* fixed_scalars = nullptr;  // q | cofactor, 64 bytes (is_torsion_free / clear_cofactor)
    u64* scratch = nullptr;
    size_t scratch_words = 0;
    u64* noalloc_arena = nullptr;  // this device's address of noalloc_scratch (hipGetSymbolAddress, once, at creation)
    int num_cus = 0;
    int reserved_cus = 0;  // CUs' worth of workgroup slots the persistent launches leave free (cheetah_ctx_set_reserved_cus)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_order = nullptr;
    std::string err;
};

static std::string g_create_err;
static std::mutex g_err_mu;

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                \
            return CHEETAH_ERR_HIP;                                                         \
        }                                                                                   \
    } while (0)

static int32_t fail(cheetah_ctx* ctx, int32_t code, const char* msg) {
    if (ctx) ctx->err = msg;
    return code;
}

static bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Generator (src/curve/affine.rs:255-275) and shift point (src/constants.rs:43-61).
static const u64 GENERATOR_XY[12] = {
    0x263a588f4b0118a1ull, 0x7757a0bcb26a142dull, 0x9215adfc1e925890ull, 0x430aad2ce14759a4ull,
    0x0534ece54de4b2c8ull, 0xb39050f01f7b1f33ull, 0xd57f0d0d47482534ull, 0x26821d894fa8ea0full,
    0xc77f564783ef13a1ull, 0x949c360784284ec2ull, 0xb7040bd639ef3cc4ull, 0x8aa635f2719d255full};
static const u64 SHIFT_XY[12] = {
    0xab95fb0d188ff2b1ull, 0xf38263a8c954493eull, 0x1bc968d6739028b7ull, 0x4f67684fe6b27d1full,
    0x6504f8bbba9077b4ull, 0x860838116d76e545ull, 0x56988918d5b02cf3ull, 0xc9fe3ef9d7652c54ull,
    0x4923e90c4d6f4538ull, 0xc747ac3f82369f27ull, 0x6e74c106972c71f5ull, 0xcf190938f1c64bb5ull};

// q (src/curve/projective.rs:886-889) and the cofactor (:875-878), little-endian, in device memory of the context.
static const u8 FQ_MODULUS_BYTES[32] = {207, 172, 212, 174, 62, 98, 67, 212, 34, 119, 21, 48, 35, 167, 122, 50,
                                        181, 55, 10, 153, 15, 191, 63, 86, 208, 34, 63, 59, 155, 89, 242, 122};
static const u8 COFACTOR_BYTES[32] = {74, 78, 48, 84, 93, 63, 188, 73, 0, 97, 239, 113, 224, 72, 11, 21, 2};


extern "C" const char* cheetah_version(void) { return "cheetah-hip 0.1 (gfx950)"; }

extern "C" const char* cheetah_last_error(const cheetah_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_create_err.c_str();
}

static int32_t create_fail(cheetah_ctx* ctx, int32_t code, const std::string& msg) {
    {
        std::lock_guard<std::mutex> lk(g_err_mu);
        g_create_err = msg;
    }
    if (ctx) cheetah_ctx_destroy(ctx);
    return code;
}

extern "C" int32_t cheetah_ctx_create(int device_id, cheetah_ctx** out) {
    if (!out) return create_fail(nullptr, CHEETAH_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return create_fail(nullptr, CHEETAH_ERR_NO_DEVICE, "no HIP device visible");
    if (device_id < 0 || device_id >= ndev) return create_fail(nullptr, CHEETAH_ERR_INVALID, "device_id out of range");
    cheetah_ctx* ctx = new (std::nothrow) cheetah_ctx();
    if (!ctx) return create_fail(nullptr, CHEETAH_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
#define CREATE_TRY(expr)                                                                             \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return create_fail(ctx, CHEETAH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
    CREATE_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return create_fail(ctx, CHEETAH_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    ctx->num_cus = prop.multiProcessorCount;
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    CREATE_TRY(hipEventCreate(&ctx->ev0));
    CREATE_TRY(hipEventCreate(&ctx->ev1));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
    CREATE_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&ctx->noalloc_arena), HIP_SYMBOL(noalloc_scratch)));
    CREATE_TRY(hipMalloc(&ctx->tables, sizeof(DeviceTables)));
    CREATE_TRY(hipMalloc(&ctx->fixed_scalars, 64));
    CREATE_TRY(hipMemcpyAsync(ctx->fixed_scalars, FQ_MODULUS_BYTES, 32, hipMemcpyHostToDevice, ctx->stream));
    CREATE_TRY(hipMemcpyAsync(ctx->fixed_scalars + 32, COFACTOR_BYTES, 32, hipMemcpyHostToDevice, ctx->stream));
    u64* seeds = nullptr;
    CREATE_TRY(hipMalloc(&seeds, 24 * sizeof(u64)));
    CREATE_TRY(hipMemcpyAsync(seeds, GENERATOR_XY, 12 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    CREATE_TRY(hipMemcpyAsync(seeds + 12, SHIFT_XY, 12 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_build_tables, dim3(1), dim3(BUILD_TABLES_THREADS), 0, ctx->stream, ctx->tables, seeds, seeds + 12);
    hipLaunchKernelGGL(k_build_wide_table, dim3(WIDE_WINDOWS), dim3(64), 0, ctx->stream, ctx->tables, seeds);
    CREATE_TRY(hipGetLastError());
    CREATE_TRY(hipStreamSynchronize(ctx->stream));
    CREATE_TRY(hipFree(seeds));
#undef CREATE_TRY
    *out = ctx;
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_ctx_destroy(cheetah_ctx* ctx) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->tables) (void)hipFree(ctx->tables);
    if (ctx->fixed_scalars) (void)hipFree(ctx->fixed_scalars);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return CHEETAH_OK;
}

// The context's kernels share its scratch arena, so its work must stay ordered when the
// stream changes: the new stream waits (on the device, no host sync) for what the
// context has already enqueued on the old one.
static int32_t switch_stream(cheetah_ctx* ctx, hipStream_t next) {
    if (next == ctx->stream) return CHEETAH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(next, ctx->ev_order, 0));
    ctx->stream = next;
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_ctx_set_stream(cheetah_ctx* ctx, void* hip_stream) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    return switch_stream(ctx, static_cast<hipStream_t>(hip_stream));
}

extern "C" int32_t cheetah_ctx_use_own_stream(cheetah_ctx* ctx) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    return switch_stream(ctx, ctx->own_stream);
}

extern "C" int32_t cheetah_ctx_synchronize(cheetah_ctx* ctx) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}

// Grid for the compute-bound per-lane kernels: enough 256-thread workgroups to
// fill every CU at the occupancy the kernel's registers allow, no more than the
// work needs.  The kernels grid-stride over the rest.
static int lane_grid(const cheetah_ctx* ctx, size_t n_lanes_needed, int blocks_per_cu) {
    size_t want = (n_lanes_needed + BLOCK - 1) / BLOCK;
    size_t cap = (size_t)(ctx->num_cus - ctx->reserved_cus) * blocks_per_cu;
    size_t g = want < cap ? want : cap;
    return g ? (int)g : 1;
}

constexpr int ROUNDS_PER_LAUNCH = 2;

static int32_t ensure_scratch(cheetah_ctx* ctx, size_t words) {
    if (ctx->scratch_words >= words) return CHEETAH_OK;
    HIP_TRY(ctx, hipDeviceSynchronize());  // nothing may still be using the old arena
    if (ctx->scratch) HIP_TRY(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_words = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->scratch, words * sizeof(u64)));
    ctx->scratch_words = words;
    return CHEETAH_OK;
}

#define CHECK_CTX(ctx)                                   \
    do {                                                 \
        if (!(ctx)) return CHEETAH_ERR_INVALID;          \
        HIP_TRY(ctx, hipSetDevice((ctx)->device));       \
    } while (0)

// ---------------------------------------------------------------- Fp6 -----

template <int OP>
static int32_t fp6_elementwise_dev(cheetah_ctx* ctx, const u64* a, const u64* b, u64* out, u8* is_some, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!a || !out || (OP == EW_MUL && !b)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned16(a) || !aligned16(out) || (b && !aligned16(b))) return fail(ctx, CHEETAH_ERR_INVALID, "Fp6 device arrays must be 16-byte aligned");
    size_t tiles = (n + BLOCK - 1) / BLOCK;
    // Measured (tools/fp6_ew_bench.py): batches beyond the Infinity Cache stream fastest with one element per
    // thread (2^24 products: 5.70 TB/s against 5.06 with a resident, grid-striding launch); batches that fit it
    // (2^20) prefer the resident grid for the shorter kernels (square: 20 us against 23).
    size_t cap = n >= ((size_t)1 << 21) ? tiles : (size_t)ctx->num_cus * 8;
    int grid = (int)(tiles < cap ? tiles : cap);
    hipLaunchKernelGGL((k_fp6_elementwise<OP>), dim3(grid), dim3(BLOCK), 0, ctx->stream, a, b, out, is_some, n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_fp6_mul_batch_dev(cheetah_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    return fp6_elementwise_dev<EW_MUL>(ctx, a, b, out, nullptr, n);
}
extern "C" int32_t cheetah_fp6_square_batch_dev(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return fp6_elementwise_dev<EW_SQUARE>(ctx, a, nullptr, out, nullptr, n);
}
extern "C" int32_t cheetah_fp6_invert_batch_dev(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n) {
    return fp6_elementwise_dev<EW_INVERT>(ctx, a, nullptr, out, is_some, n);
}
extern "C" int32_t cheetah_fp6_sqrt_batch_dev(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n) {
    return fp6_elementwise_dev<EW_SQRT>(ctx, a, nullptr, out, is_some, n);
}

// ------------------------------------------------------ batch_normalize -----

template <int JAC>
static int32_t batch_normalize_dev(cheetah_ctx* ctx, const u64* xyz, u64* xy, u8* inf, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!xyz || !xy || !inf) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned16(xyz) || !aligned16(xy)) return fail(ctx, CHEETAH_ERR_INVALID, "point arrays must be 16-byte aligned");
    constexpr int CHUNK = 8;
    size_t nchunks = (n + CHUNK - 1) / CHUNK;
    int grid = lane_grid(ctx, nchunks, CH_BN_WAVES);
    hipLaunchKernelGGL((k_batch_normalize<JAC, CHUNK>), dim3(grid), dim3(BLOCK), 0, ctx->stream, xyz, xy, inf, n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_batch_normalize_projective_dev(cheetah_ctx* ctx, const uint64_t* xyz, uint64_t* xy, uint8_t* inf, size_t n) {
    return batch_normalize_dev<0>(ctx, xyz, xy, inf, n);
}
extern "C" int32_t cheetah_batch_normalize_jacobian_dev(cheetah_ctx* ctx, const uint64_t* xyz, uint64_t* xy, uint8_t* inf, size_t n) {
    return batch_normalize_dev<1>(ctx, xyz, xy, inf, n);
}

// ------------------------------------------------------------- ladders -----

// Which reference ladder a (in_form, out_form) pair names:
//   PROJECTIVE on either side -> ProjectivePoint::multiply*           (projective accumulator)
//   otherwise                 -> JacobianPoint / AffinePoint::multiply* (modified-Jacobian accumulator;
//                                AffinePoint's methods delegate to JacobianPoint's, src/curve/affine.rs:373-386)
// PROJECTIVE <-> JACOBIAN has no counterpart in the reference and is rejected.
static int32_t check_forms(cheetah_ctx* ctx, int in_form, int out_form) {
    if (in_form < CHEETAH_AFFINE || in_form > CHEETAH_JACOBIAN) return fail(ctx, CHEETAH_ERR_INVALID, "in_form must be AFFINE, PROJECTIVE or JACOBIAN");
    if (out_form < CHEETAH_AFFINE || out_form > CHEETAH_JACOBIAN) return fail(ctx, CHEETAH_ERR_INVALID, "out_form must be AFFINE, PROJECTIVE or JACOBIAN");
    if ((in_form == CHEETAH_PROJECTIVE && out_form == CHEETAH_JACOBIAN) || (in_form == CHEETAH_JACOBIAN && out_form == CHEETAH_PROJECTIVE))
        return fail(ctx, CHEETAH_ERR_INVALID, "PROJECTIVE and JACOBIAN cannot be mixed in one call");
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_mul_by_generator_batch_dev(cheetah_ctx* ctx, const uint8_t* scalars32, uint64_t* out, uint8_t* inf,
                                                      size_t n, int out_form) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, CHEETAH_AFFINE, out_form)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!scalars32 || !out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned8(scalars32) || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
    if (out_form == CHEETAH_AFFINE) {  // affine results: the widened comb (one 1 024-thread workgroup per CU)
        size_t want = (n + WIDE_BLOCK - 1) / WIDE_BLOCK;
        const size_t cus_w = (size_t)(ctx->num_cus - ctx->reserved_cus);
        int grid_w = (int)(want < cus_w ? want : cus_w);
        const size_t per_launch_w = (size_t)grid_w * WIDE_BLOCK * ROUNDS_PER_LAUNCH * 4;
        for (size_t off = 0; off < n; off += per_launch_w) {
            const size_t cnt = n - off < per_launch_w ? n - off : per_launch_w;
            hipLaunchKernelGGL(k_mul_by_generator_wide, dim3(grid_w), dim3(WIDE_BLOCK), 0, ctx->stream, ctx->tables,
                               scalars32 + off * 32, out + off * 12, inf + off, cnt);
        }
        HIP_TRY(ctx, hipGetLastError());
        return CHEETAH_OK;
    }
    int grid = lane_grid(ctx, n, CH_GEN_WAVES);
    const size_t out_w = out_form == CHEETAH_AFFINE ? 12 : 18;
    const size_t per_launch = (size_t)grid * BLOCK * ROUNDS_PER_LAUNCH * 4;  // bounded launches, as in multiply_dev
    for (size_t off = 0; off < n; off += per_launch) {
        const size_t cnt = n - off < per_launch ? n - off : per_launch;
        const u8* sc_o = scalars32 + off * 32;
        u64* out_o = out + off * out_w;
        u8* inf_o = inf ? inf + off : nullptr;
        if (out_form == CHEETAH_AFFINE)
            hipLaunchKernelGGL((k_mul_by_generator<0>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, sc_o, out_o, inf_o, cnt);
        else if (out_form == CHEETAH_PROJECTIVE)
            hipLaunchKernelGGL((k_mul_by_generator<1>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, sc_o, out_o, inf_o, cnt);
        else
            hipLaunchKernelGGL((k_mul_by_generator_lookup<2>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, sc_o, out_o, inf_o, cnt);
    }
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

// arena == nullptr: the context's own scratch (grown on demand); otherwise a caller-chosen arena that holds
// max_blocks workgroups' tables, and the grids are capped to it (the kernels grid-stride over the rest).
template <bool VARTIME>
static int32_t multiply_dev(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* scalars32, u64* out, u8* inf, size_t n,
                            int out_form, u64* arena = nullptr, int max_blocks = 0) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!pts || !scalars32 || !out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned8(pts) || !aligned8(scalars32) || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
    // multiply_vartime with AFFINE results: the fixed-window ladder with indexed lookups (k_multiply_ct<..., DIRECT>)
    // for every wave whose scalars are all below 2^255, then the reference's NAF ladder for the remaining waves.
    const bool direct = VARTIME && out_form == CHEETAH_AFFINE;
    // the ONLY_WIDE instance of the NAF kernel (normally nothing to do) is compiled for 2 waves: no scratch
    int grid_vt = lane_grid(ctx, n, direct ? 2 : CH_VT_WAVES), grid_ct = lane_grid(ctx, n, CH_CT_WAVES);
    if (arena) {
        if (grid_vt > max_blocks) grid_vt = max_blocks;
        if (grid_ct > max_blocks) grid_ct = max_blocks;
    } else {
        if (int32_t rc = ensure_scratch(ctx, (size_t)(grid_vt > grid_ct ? grid_vt : grid_ct) * LaneScratch::WORDS_PER_BLOCK)) return rc;
        arena = ctx->scratch;
    }
    const int grid = (VARTIME && !direct) ? grid_vt : grid_ct;
    // One launch covers at most ROUNDS_PER_LAUNCH rounds of the resident grid (~25 ms of
    // work).  The workgroups are persistent within a launch and fill every CU, so a kernel
    // on another stream (the RCCL all-gather of the previous chunk) can only get CU slots
    // at a launch boundary: bounding the launch bounds how long it waits.
    const size_t in_w = in_form == CHEETAH_AFFINE ? 12 : 18, out_w = out_form == CHEETAH_AFFINE ? 12 : 18;
    const size_t per_launch = (size_t)grid * BLOCK * ROUNDS_PER_LAUNCH;
    for (size_t off = 0; off < n; off += per_launch) {
        const size_t cnt = n - off < per_launch ? n - off : per_launch;
        const u64* pts_o = pts + off * in_w;
        const u8* sc_o = scalars32 + off * 32;
        u64* out_o = out + off * out_w;
        u8* inf_o = inf ? inf + off : nullptr;
#define LAUNCH(KERN, A, I, O)                                                                                                   \
    do {                                                                                                                       \
        if constexpr (VARTIME && O == 0) {                                                                                             \
            hipLaunchKernelGGL((k_multiply_ct<A, I, O, true>), dim3(grid_ct), dim3(BLOCK), 0, ctx->stream, ctx->tables, pts_o,  \
                               sc_o, out_o, inf_o, cnt, arena);                                                         \
            hipLaunchKernelGGL((k_multiply_vartime<A, I, O, true>), dim3(grid_vt), dim3(BLOCK), 0, ctx->stream, ctx->tables,    \
                               pts_o, sc_o, out_o, inf_o, cnt, arena, 0);                                               \
        } else if constexpr (VARTIME)                                                                                              \
            hipLaunchKernelGGL((k_multiply_vartime<A, I, O, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables,      \
                               pts_o, sc_o, out_o, inf_o, cnt, arena, 0);                                               \
        else                                                                                                                   \
            hipLaunchKernelGGL((k_multiply_ct<A, I, O, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, pts_o,    \
                               sc_o, out_o, inf_o, cnt, arena);                                                         \
    } while (0)
#define DISPATCH(KERN)                                         \
    switch (in_form * 3 + out_form) {                          \
        case 0: LAUNCH(KERN, MJac, 0, 0); break; /* A -> A */ \
        case 1: LAUNCH(KERN, Proj, 0, 1); break; /* A -> P */ \
        case 2: LAUNCH(KERN, MJac, 0, 2); break; /* A -> J */ \
        case 3: LAUNCH(KERN, MJac, 1, 0); break; /* P -> A: the affine point is the same on either accumulator; MJac is cheaper */ \
        case 4: LAUNCH(KERN, Proj, 1, 1); break; /* P -> P */ \
        case 6: LAUNCH(KERN, MJac, 2, 0); break; /* J -> A */ \
        default: LAUNCH(KERN, MJac, 2, 2); break; /* J -> J */ \
    }
        DISPATCH(_)
    }
#undef DISPATCH
#undef LAUNCH
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_multiply_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32,
                                              uint64_t* out, uint8_t* inf, size_t n, int out_form) {
    return multiply_dev<false>(ctx, pts, in_form, scalars32, out, inf, n, out_form);
}
extern "C" int32_t cheetah_multiply_vartime_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form,
                                                      const uint8_t* scalars32, uint64_t* out, uint8_t* inf, size_t n,
                                                      int out_form) {
    return multiply_dev<true>(ctx, pts, in_form, scalars32, out, inf, n, out_form);
}
// cheetah_multiply_batch_dev for callers that cannot allocate: the same kernels on the file-scope arena.
// Calls no hipMalloc / hipFree (nor anything else that allocates): argument checks, hipSetDevice, kernel launches.
// The arena needs no clearing: a workgroup writes every table entry it later reads, in every round.
extern "C" int32_t cheetah_multiply_batch_dev_noalloc(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32,
                                                      uint64_t* out, uint8_t* inf, size_t n, int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (n > (size_t)CHEETAH_NOALLOC_MAX_N)
        return fail(ctx, CHEETAH_ERR_INVALID, "cheetah_multiply_batch_dev_noalloc: n exceeds CHEETAH_NOALLOC_MAX_N (use cheetah_multiply_batch_dev)");
    if (!ctx->noalloc_arena) return fail(ctx, CHEETAH_ERR_HIP, "the static arena's device address is not known");
    return multiply_dev<false>(ctx, pts, in_form, scalars32, out, inf, n, out_form, ctx->noalloc_arena, NOALLOC_MAX_BLOCKS);
}

// ------------------------------------------------- host-pointer wrappers -----
// Copy in, run the same device entry point, copy out.  PCIe-inclusive; the
// `_dev` forms are what the benchmarks time.

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};
}  // namespace

#define H2D(ctx, dst, src, bytes) HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (ctx)->stream))
#define D2H(ctx, dst, src, bytes) HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (ctx)->stream))

extern "C" int32_t cheetah_fp6_mul_batch(cheetah_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!a || !b || !out) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    DevBuf da, db, dout;
    HIP_TRY(ctx, da.alloc(n * 48));
    HIP_TRY(ctx, db.alloc(n * 48));
    HIP_TRY(ctx, dout.alloc(n * 48));
    H2D(ctx, da.p, a, n * 48);
    H2D(ctx, db.p, b, n * 48);
    if (int32_t rc = cheetah_fp6_mul_batch_dev(ctx, da.as<u64>(), db.as<u64>(), dout.as<u64>(), n)) return rc;
    D2H(ctx, out, dout.p, n * 48);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_fp6_square_batch(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!a || !out) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    DevBuf da, dout;
    HIP_TRY(ctx, da.alloc(n * 48));
    HIP_TRY(ctx, dout.alloc(n * 48));
    H2D(ctx, da.p, a, n * 48);
    if (int32_t rc = cheetah_fp6_square_batch_dev(ctx, da.as<u64>(), dout.as<u64>(), n)) return rc;
    D2H(ctx, out, dout.p, n * 48);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
template <int OP>
static int32_t fp6_option_host(cheetah_ctx* ctx, const u64* a, u64* out, u8* is_some, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!a || !out || !is_some) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    DevBuf da, dout, dsome;
    HIP_TRY(ctx, da.alloc(n * 48));
    HIP_TRY(ctx, dout.alloc(n * 48));
    HIP_TRY(ctx, dsome.alloc(n));
    H2D(ctx, da.p, a, n * 48);
    if (int32_t rc = fp6_elementwise_dev<OP>(ctx, da.as<u64>(), nullptr, dout.as<u64>(), dsome.as<u8>(), n)) return rc;
    D2H(ctx, out, dout.p, n * 48);
    D2H(ctx, is_some, dsome.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_fp6_invert_batch(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n) {
    return fp6_option_host<EW_INVERT>(ctx, a, out, is_some, n);
}
extern "C" int32_t cheetah_fp6_sqrt_batch(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n) {
    return fp6_option_host<EW_SQRT>(ctx, a, out, is_some, n);
}

template <int JAC>
static int32_t batch_normalize_host(cheetah_ctx* ctx, const u64* xyz, u64* xy, u8* inf, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!xyz || !xy || !inf) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    DevBuf din, dxy, dinf;
    HIP_TRY(ctx, din.alloc(n * 144));
    HIP_TRY(ctx, dxy.alloc(n * 96));
    HIP_TRY(ctx, dinf.alloc(n));
    H2D(ctx, din.p, xyz, n * 144);
    if (int32_t rc = batch_normalize_dev<JAC>(ctx, din.as<u64>(), dxy.as<u64>(), dinf.as<u8>(), n)) return rc;
    D2H(ctx, xy, dxy.p, n * 96);
    D2H(ctx, inf, dinf.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_batch_normalize_projective(cheetah_ctx* ctx, const uint64_t* xyz, uint64_t* xy, uint8_t* inf, size_t n) {
    return batch_normalize_host<0>(ctx, xyz, xy, inf, n);
}
extern "C" int32_t cheetah_batch_normalize_jacobian(cheetah_ctx* ctx, const uint64_t* xyz, uint64_t* xy, uint8_t* inf, size_t n) {
    return batch_normalize_host<1>(ctx, xyz, xy, inf, n);
}

extern "C" int32_t cheetah_mul_by_generator_batch(cheetah_ctx* ctx, const uint8_t* scalars32, uint64_t* out, uint8_t* inf, size_t n,
                                                  int out_form) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, CHEETAH_AFFINE, out_form)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!scalars32 || !out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (out_form != CHEETAH_JACOBIAN)  // BASEPOINT_LOOKUP's LookupTable<128> covers every radix-256 digit
        for (size_t i = 0; i < n; i++)
            if ((scalars32[32 * i + 31] >> 4) > 7)
                return fail(ctx, CHEETAH_ERR_SCALAR, "scalar with top nibble > 7: outside BasePointTable's LookupTable<8> range");
    const size_t w = out_form == CHEETAH_AFFINE ? 96 : 144;  // projective and Jacobian: x | y | z
    DevBuf dsc, dout, dinf;
    HIP_TRY(ctx, dsc.alloc(n * 32));
    HIP_TRY(ctx, dout.alloc(n * w));
    HIP_TRY(ctx, dinf.alloc(n));
    H2D(ctx, dsc.p, scalars32, n * 32);
    if (int32_t rc = cheetah_mul_by_generator_batch_dev(ctx, dsc.as<u8>(), dout.as<u64>(), dinf.as<u8>(), n, out_form)) return rc;
    D2H(ctx, out, dout.p, n * w);
    if (inf) D2H(ctx, inf, dinf.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}

template <bool VARTIME>
static int32_t multiply_host(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* scalars32, u64* out, u8* inf, size_t n,
                             int out_form) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!pts || !scalars32 || !out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    const size_t wi = in_form == CHEETAH_AFFINE ? 96 : 144, wo = out_form == CHEETAH_AFFINE ? 96 : 144;
    DevBuf dp, dsc, dout, dinf;
    HIP_TRY(ctx, dp.alloc(n * wi));
    HIP_TRY(ctx, dsc.alloc(n * 32));
    HIP_TRY(ctx, dout.alloc(n * wo));
    HIP_TRY(ctx, dinf.alloc(n));
    H2D(ctx, dp.p, pts, n * wi);
    H2D(ctx, dsc.p, scalars32, n * 32);
    if (int32_t rc = multiply_dev<VARTIME>(ctx, dp.as<u64>(), in_form, dsc.as<u8>(), dout.as<u64>(), dinf.as<u8>(), n, out_form)) return rc;
    D2H(ctx, out, dout.p, n * wo);
    if (inf) D2H(ctx, inf, dinf.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_multiply_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32, uint64_t* out,
                                          uint8_t* inf, size_t n, int out_form) {
    return multiply_host<false>(ctx, pts, in_form, scalars32, out, inf, n, out_form);
}
extern "C" int32_t cheetah_multiply_vartime_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32,
                                                  uint64_t* out, uint8_t* inf, size_t n, int out_form) {
    return multiply_host<true>(ctx, pts, in_form, scalars32, out, inf, n, out_form);
}


// --------------------------------------------- multi-scalar ladders, checks -----

template <bool VARTIME>
static int32_t multiply_many_dev(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* scalars32, int m, int with_basepoint,
                                 u64* out, u8* inf, size_t n_groups, int out_form) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    if (m < 1 || m > CHEETAH_MAX_MANY) return fail(ctx, CHEETAH_ERR_INVALID, "m must be in [1, CHEETAH_MAX_MANY]");
    if (n_groups == 0) return CHEETAH_OK;
    if (!pts || !scalars32 || !out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned8(pts) || !aligned8(scalars32) || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
    const int ns = m + (with_basepoint ? 1 : 0);
    int grid = lane_grid(ctx, n_groups, CH_MANY_WAVES);
    const size_t wpb = ManyScratch::words(m, ns);
    if (int32_t rc = ensure_scratch(ctx, (size_t)grid * wpb)) return rc;
    const size_t in_w = in_form == CHEETAH_AFFINE ? 12 : 18, out_w = out_form == CHEETAH_AFFINE ? 12 : 18;
    size_t rounds = ROUNDS_PER_LAUNCH / (size_t)m;
    if (rounds == 0) rounds = 1;
    const size_t per_launch = (size_t)grid * BLOCK * rounds;
    for (size_t off = 0; off < n_groups; off += per_launch) {
        const size_t cnt = n_groups - off < per_launch ? n_groups - off : per_launch;
        const u64* pts_o = pts + off * m * in_w;
        const u8* sc_o = scalars32 + off * ns * 32;
        u64* out_o = out + off * out_w;
        u8* inf_o = inf ? inf + off : nullptr;
#define LAUNCH(A, I, O)                                                                                                           \
    do {                                                                                                                          \
        if (VARTIME)                                                                                                              \
            hipLaunchKernelGGL((k_multiply_many_vartime<A, I, O>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, pts_o,    \
                               sc_o, out_o, inf_o, cnt, m, with_basepoint, ctx->scratch, wpb);                                    \
        else                                                                                                                      \
            hipLaunchKernelGGL((k_multiply_many_ct<A, I, O>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, pts_o, sc_o,   \
                               out_o, inf_o, cnt, m, ctx->scratch, wpb);                                                          \
    } while (0)
        switch (in_form * 3 + out_form) {
            case 0: LAUNCH(MJac, 0, 0); break;
            case 1: LAUNCH(Proj, 0, 1); break;
            case 2: LAUNCH(MJac, 0, 2); break;
            case 3: LAUNCH(Proj, 1, 0); break;
            case 4: LAUNCH(Proj, 1, 1); break;
            case 6: LAUNCH(MJac, 2, 0); break;
            default: LAUNCH(MJac, 2, 2); break;
        }
#undef LAUNCH
    }
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_multiply_many_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32, int m,
                                                   uint64_t* out, uint8_t* inf, size_t n_groups, int out_form) {
    return multiply_many_dev<false>(ctx, pts, in_form, scalars32, m, 0, out, inf, n_groups, out_form);
}
extern "C" int32_t cheetah_multiply_many_vartime_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form,
                                                           const uint8_t* scalars32, int m, uint64_t* out, uint8_t* inf,
                                                           size_t n_groups, int out_form) {
    return multiply_many_dev<true>(ctx, pts, in_form, scalars32, m, 0, out, inf, n_groups, out_form);
}
extern "C" int32_t cheetah_multiply_double_with_basepoint_vartime_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form,
                                                                            const uint8_t* scalars64, uint64_t* out, uint8_t* inf,
                                                                            size_t n, int out_form) {
    return multiply_many_dev<true>(ctx, pts, in_form, scalars64, 1, 1, out, inf, n, out_form);
}

// One fixed scalar for the whole batch through the vartime ladder (every lane has the
// same NAF, so the wave never diverges).  out_form 3 = identity flag only.
static int32_t fixed_scalar_dev(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* dev_scalar, u64* out, u8* inf, size_t n,
                                int out_form) {
    int grid = lane_grid(ctx, n, CH_VT_WAVES);
    if (int32_t rc = ensure_scratch(ctx, (size_t)grid * LaneScratch::WORDS_PER_BLOCK)) return rc;
    const size_t in_w = in_form == CHEETAH_AFFINE ? 12 : 18, out_w = out_form == CHEETAH_AFFINE ? 12 : 18;
    const size_t per_launch = (size_t)grid * BLOCK * ROUNDS_PER_LAUNCH;
    const bool proj = in_form == CHEETAH_PROJECTIVE || out_form == CHEETAH_PROJECTIVE;
    for (size_t off = 0; off < n; off += per_launch) {
        const size_t cnt = n - off < per_launch ? n - off : per_launch;
        const u64* pts_o = pts + off * in_w;
        u64* out_o = out ? out + off * out_w : nullptr;
        u8* inf_o = inf ? inf + off : nullptr;
#define LAUNCH(A, I, O) \
    hipLaunchKernelGGL((k_multiply_vartime<A, I, O>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, pts_o, dev_scalar, out_o, inf_o, cnt, ctx->scratch, 1)
        if (out_form == 3) {  // flag only
            if (in_form == CHEETAH_PROJECTIVE) LAUNCH(Proj, 1, 3);
            else if (in_form == CHEETAH_JACOBIAN) LAUNCH(MJac, 2, 3);
            else LAUNCH(MJac, 0, 3);
        } else if (out_form == 4) {  // flags[i] &= is_identity (affine input)
            LAUNCH(MJac, 0, 4);
        } else if (proj) {
            if (in_form == CHEETAH_AFFINE) LAUNCH(Proj, 0, 1);
            else if (out_form == CHEETAH_AFFINE) LAUNCH(Proj, 1, 0);
            else LAUNCH(Proj, 1, 1);
        } else {
            if (in_form == CHEETAH_AFFINE && out_form == CHEETAH_AFFINE) LAUNCH(MJac, 0, 0);
            else if (in_form == CHEETAH_AFFINE) LAUNCH(MJac, 0, 2);
            else if (out_form == CHEETAH_AFFINE) LAUNCH(MJac, 2, 0);
            else LAUNCH(MJac, 2, 2);
        }
#undef LAUNCH
    }
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_is_torsion_free_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint8_t* flags, size_t n) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, in_form, CHEETAH_AFFINE)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!pts || !flags || !aligned8(pts)) return fail(ctx, CHEETAH_ERR_INVALID, "bad pointer");
    return fixed_scalar_dev(ctx, pts, in_form, ctx->fixed_scalars, nullptr, flags, n, 3);
}
extern "C" int32_t cheetah_clear_cofactor_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint64_t* out, uint8_t* inf,
                                                    size_t n, int out_form) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!pts || !out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned8(pts) || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
    return fixed_scalar_dev(ctx, pts, in_form, ctx->fixed_scalars + 32, out, inf, n, out_form);
}

// host-pointer forms
template <bool VARTIME>
static int32_t multiply_many_host(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* scalars32, int m, int with_basepoint,
                                  u64* out, u8* inf, size_t n_groups, int out_form) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    if (m < 1 || m > CHEETAH_MAX_MANY) return fail(ctx, CHEETAH_ERR_INVALID, "m must be in [1, CHEETAH_MAX_MANY]");
    if (n_groups == 0) return CHEETAH_OK;
    if (!pts || !scalars32 || !out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    const size_t wi = in_form == CHEETAH_AFFINE ? 96 : 144, wo = out_form == CHEETAH_AFFINE ? 96 : 144;
    const size_t ns = (size_t)m + (with_basepoint ? 1 : 0);
    DevBuf dp, dsc, dout, dinf;
    HIP_TRY(ctx, dp.alloc(n_groups * m * wi));
    HIP_TRY(ctx, dsc.alloc(n_groups * ns * 32));
    HIP_TRY(ctx, dout.alloc(n_groups * wo));
    HIP_TRY(ctx, dinf.alloc(n_groups));
    H2D(ctx, dp.p, pts, n_groups * m * wi);
    H2D(ctx, dsc.p, scalars32, n_groups * ns * 32);
    if (int32_t rc = multiply_many_dev<VARTIME>(ctx, dp.as<u64>(), in_form, dsc.as<u8>(), m, with_basepoint, dout.as<u64>(),
                                                dinf.as<u8>(), n_groups, out_form))
        return rc;
    D2H(ctx, out, dout.p, n_groups * wo);
    if (inf) D2H(ctx, inf, dinf.p, n_groups);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_multiply_many_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32, int m,
                                               uint64_t* out, uint8_t* inf, size_t n_groups, int out_form) {
    return multiply_many_host<false>(ctx, pts, in_form, scalars32, m, 0, out, inf, n_groups, out_form);
}
extern "C" int32_t cheetah_multiply_many_vartime_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32,
                                                       int m, uint64_t* out, uint8_t* inf, size_t n_groups, int out_form) {
    return multiply_many_host<true>(ctx, pts, in_form, scalars32, m, 0, out, inf, n_groups, out_form);
}
extern "C" int32_t cheetah_multiply_double_with_basepoint_vartime_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form,
                                                                        const uint8_t* scalars64, uint64_t* out, uint8_t* inf,
                                                                        size_t n, int out_form) {
    return multiply_many_host<true>(ctx, pts, in_form, scalars64, 1, 1, out, inf, n, out_form);
}
extern "C" int32_t cheetah_is_torsion_free_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint8_t* flags, size_t n) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, in_form, CHEETAH_AFFINE)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!pts || !flags) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    const size_t wi = in_form == CHEETAH_AFFINE ? 96 : 144;
    DevBuf dp, df;
    HIP_TRY(ctx, dp.alloc(n * wi));
    HIP_TRY(ctx, df.alloc(n));
    H2D(ctx, dp.p, pts, n * wi);
    if (int32_t rc = cheetah_is_torsion_free_batch_dev(ctx, dp.as<u64>(), in_form, df.as<u8>(), n)) return rc;
    D2H(ctx, flags, df.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_clear_cofactor_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint64_t* out, uint8_t* inf,
                                                size_t n, int out_form) {
    CHECK_CTX(ctx);
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!pts || !out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    const size_t wi = in_form == CHEETAH_AFFINE ? 96 : 144, wo = out_form == CHEETAH_AFFINE ? 96 : 144;
    DevBuf dp, dout, dinf;
    HIP_TRY(ctx, dp.alloc(n * wi));
    HIP_TRY(ctx, dout.alloc(n * wo));
    HIP_TRY(ctx, dinf.alloc(n));
    H2D(ctx, dp.p, pts, n * wi);
    if (int32_t rc = cheetah_clear_cofactor_batch_dev(ctx, dp.as<u64>(), in_form, dout.as<u64>(), dinf.as<u8>(), n, out_form)) return rc;
    D2H(ctx, out, dout.p, n * wo);
    if (inf) D2H(ctx, inf, dinf.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}


// -------------------------------------------------------------- encodings -----

static bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

template <int REC>
static int32_t encode_dev(cheetah_ctx* ctx, const u64* xy, const u8* inf, u8* out, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!xy || !inf || !out) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned16(xy) || !aligned4(out)) return fail(ctx, CHEETAH_ERR_INVALID, "xy must be 16-byte aligned, the byte array 4-byte aligned");
    size_t tiles = (n + BLOCK - 1) / BLOCK, cap = (size_t)ctx->num_cus * 8;
    hipLaunchKernelGGL((k_encode<REC>), dim3((int)(tiles < cap ? tiles : cap)), dim3(BLOCK), 0, ctx->stream, xy, inf, out, n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_encode_uncompressed_batch_dev(cheetah_ctx* ctx, const uint64_t* xy, const uint8_t* inf, uint8_t* out97, size_t n) {
    return encode_dev<97>(ctx, xy, inf, out97, n);
}
extern "C" int32_t cheetah_encode_compressed_batch_dev(cheetah_ctx* ctx, const uint64_t* xy, const uint8_t* inf, uint8_t* out49, size_t n) {
    return encode_dev<49>(ctx, xy, inf, out49, n);
}
extern "C" int32_t cheetah_decode_uncompressed_batch_dev(cheetah_ctx* ctx, const uint8_t* in97, uint64_t* xy, uint8_t* inf,
                                                         uint8_t* is_some, size_t n, int checked) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!in97 || !xy || !inf || !is_some) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned16(xy) || !aligned4(in97)) return fail(ctx, CHEETAH_ERR_INVALID, "xy must be 16-byte aligned, the byte array 4-byte aligned");
    size_t tiles = (n + BLOCK - 1) / BLOCK, cap = (size_t)ctx->num_cus * 8;
    int grid = (int)(tiles < cap ? tiles : cap);
    if (checked) {
        hipLaunchKernelGGL((k_decode_uncompressed<1>), dim3(grid), dim3(BLOCK), 0, ctx->stream, in97, xy, inf, is_some, n);
        HIP_TRY(ctx, hipGetLastError());
        if (int32_t rc = fixed_scalar_dev(ctx, xy, CHEETAH_AFFINE, ctx->fixed_scalars, nullptr, is_some, n, 4)) return rc;  // is_some &= is_torsion_free
        hipLaunchKernelGGL(k_zero_rejected, dim3(grid), dim3(BLOCK), 0, ctx->stream, xy, inf, is_some, n);
        HIP_TRY(ctx, hipGetLastError());
        return CHEETAH_OK;
    }
    hipLaunchKernelGGL((k_decode_uncompressed<0>), dim3(grid), dim3(BLOCK), 0, ctx->stream, in97, xy, inf, is_some, n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_decode_compressed_batch_dev(cheetah_ctx* ctx, const uint8_t* in49, uint64_t* xy, uint8_t* inf,
                                                       uint8_t* is_some, size_t n, int checked) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!in49 || !xy || !inf || !is_some) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned16(xy) || !aligned4(in49)) return fail(ctx, CHEETAH_ERR_INVALID, "xy must be 16-byte aligned, the byte array 4-byte aligned");
    size_t tiles = (n + BLOCK - 1) / BLOCK, cap = (size_t)ctx->num_cus * 8;
    int grid = (int)(tiles < cap ? tiles : cap);
    hipLaunchKernelGGL(k_decode_compressed, dim3(grid), dim3(BLOCK), 0, ctx->stream, in49, xy, inf, is_some, n);
    HIP_TRY(ctx, hipGetLastError());
    if (checked) {
        if (int32_t rc = fixed_scalar_dev(ctx, xy, CHEETAH_AFFINE, ctx->fixed_scalars, nullptr, is_some, n, 4)) return rc;  // is_some &= is_torsion_free
        hipLaunchKernelGGL(k_zero_rejected, dim3(grid), dim3(BLOCK), 0, ctx->stream, xy, inf, is_some, n);
        HIP_TRY(ctx, hipGetLastError());
    }
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_affine_from_x_batch_dev(cheetah_ctx* ctx, const uint64_t* x, uint64_t* xy, uint8_t* inf, uint8_t* is_some,
                                                   size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!x || !xy || !inf || !is_some) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned16(x) || !aligned16(xy)) return fail(ctx, CHEETAH_ERR_INVALID, "Fp6 device arrays must be 16-byte aligned");
    size_t tiles = (n + BLOCK - 1) / BLOCK, cap = (size_t)ctx->num_cus * 8;
    hipLaunchKernelGGL(k_affine_from_x, dim3((int)(tiles < cap ? tiles : cap)), dim3(BLOCK), 0, ctx->stream, x, xy, inf, is_some, n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

template <int REC>
static int32_t encode_host(cheetah_ctx* ctx, const u64* xy, const u8* inf, u8* out, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!xy || !inf || !out) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    DevBuf dxy, dinf, dout;
    HIP_TRY(ctx, dxy.alloc(n * 96));
    HIP_TRY(ctx, dinf.alloc(n));
    HIP_TRY(ctx, dout.alloc(n * REC));
    H2D(ctx, dxy.p, xy, n * 96);
    H2D(ctx, dinf.p, inf, n);
    if (int32_t rc = encode_dev<REC>(ctx, dxy.as<u64>(), dinf.as<u8>(), dout.as<u8>(), n)) return rc;
    D2H(ctx, out, dout.p, n * REC);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_encode_uncompressed_batch(cheetah_ctx* ctx, const uint64_t* xy, const uint8_t* inf, uint8_t* out97, size_t n) {
    return encode_host<97>(ctx, xy, inf, out97, n);
}
extern "C" int32_t cheetah_encode_compressed_batch(cheetah_ctx* ctx, const uint64_t* xy, const uint8_t* inf, uint8_t* out49, size_t n) {
    return encode_host<49>(ctx, xy, inf, out49, n);
}
extern "C" int32_t cheetah_decode_uncompressed_batch(cheetah_ctx* ctx, const uint8_t* in97, uint64_t* xy, uint8_t* inf, uint8_t* is_some,
                                                     size_t n, int checked) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!in97 || !xy || !inf || !is_some) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    DevBuf din, dxy, dinf, dsome;
    HIP_TRY(ctx, din.alloc(n * 97));
    HIP_TRY(ctx, dxy.alloc(n * 96));
    HIP_TRY(ctx, dinf.alloc(n));
    HIP_TRY(ctx, dsome.alloc(n));
    H2D(ctx, din.p, in97, n * 97);
    if (int32_t rc = cheetah_decode_uncompressed_batch_dev(ctx, din.as<u8>(), dxy.as<u64>(), dinf.as<u8>(), dsome.as<u8>(), n, checked)) return rc;
    D2H(ctx, xy, dxy.p, n * 96);
    D2H(ctx, inf, dinf.p, n);
    D2H(ctx, is_some, dsome.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_decode_compressed_batch(cheetah_ctx* ctx, const uint8_t* in49, uint64_t* xy, uint8_t* inf, uint8_t* is_some,
                                                   size_t n, int checked) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!in49 || !xy || !inf || !is_some) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    DevBuf din, dxy, dinf, dsome;
    HIP_TRY(ctx, din.alloc(n * 49));
    HIP_TRY(ctx, dxy.alloc(n * 96));
    HIP_TRY(ctx, dinf.alloc(n));
    HIP_TRY(ctx, dsome.alloc(n));
    H2D(ctx, din.p, in49, n * 49);
    if (int32_t rc = cheetah_decode_compressed_batch_dev(ctx, din.as<u8>(), dxy.as<u64>(), dinf.as<u8>(), dsome.as<u8>(), n, checked)) return rc;
    D2H(ctx, xy, dxy.p, n * 96);
    D2H(ctx, inf, dinf.p, n);
    D2H(ctx, is_some, dsome.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_affine_from_x_batch(cheetah_ctx* ctx, const uint64_t* x, uint64_t* xy, uint8_t* inf, uint8_t* is_some, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!x || !xy || !inf || !is_some) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    DevBuf dx, dxy, dinf, dsome;
    HIP_TRY(ctx, dx.alloc(n * 48));
    HIP_TRY(ctx, dxy.alloc(n * 96));
    HIP_TRY(ctx, dinf.alloc(n));
    HIP_TRY(ctx, dsome.alloc(n));
    H2D(ctx, dx.p, x, n * 48);
    if (int32_t rc = cheetah_affine_from_x_batch_dev(ctx, dx.as<u64>(), dxy.as<u64>(), dinf.as<u8>(), dsome.as<u8>(), n)) return rc;
    D2H(ctx, xy, dxy.p, n * 96);
    D2H(ctx, inf, dinf.p, n);
    D2H(ctx, is_some, dsome.p, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}

// ------------------------------------------------------- tables, synth -----

extern "C" int32_t cheetah_read_table(cheetah_ctx* ctx, int which, uint64_t* out_xy, size_t max_points) {
    CHECK_CTX(ctx);
    if (!out_xy) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    const char* base = reinterpret_cast<const char*>(ctx->tables);
    size_t off, pts;
    if (which == 0) {
        off = offsetof(DeviceTables, basepoint);
        pts = 256;
    } else if (which == 1) {
        off = offsetof(DeviceTables, shift);
        pts = 3;
    } else if (which == 2) {
        off = offsetof(DeviceTables, basepoint_lookup);
        pts = 128;
    } else if (which == 3) {
        off = offsetof(DeviceTables, odd_multiples);
        pts = 64;
    } else {
        return fail(ctx, CHEETAH_ERR_INVALID, "unknown table");
    }
    if (pts > max_points) pts = max_points;
    HIP_TRY(ctx, hipMemcpyAsync(out_xy, base + off, pts * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_synth_fp6_dev(cheetah_ctx* ctx, uint64_t seed, uint64_t start, uint64_t* out, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!out || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "bad pointer");
    hipLaunchKernelGGL(k_synth_fp6, dim3(ctx->num_cus * 8), dim3(BLOCK), 0, ctx->stream, seed, start, out, n * 6);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_synth_scalars_dev(cheetah_ctx* ctx, uint64_t seed, uint64_t start, uint8_t* out32, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!out32 || !aligned8(out32)) return fail(ctx, CHEETAH_ERR_INVALID, "bad pointer");
    hipLaunchKernelGGL(k_synth_scalars, dim3(ctx->num_cus * 8), dim3(BLOCK), 0, ctx->stream, seed, start,
                       reinterpret_cast<u64*>(out32), n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

// ---------------------------------------------------------- diagnostics -----

extern "C" int32_t cheetah_debug_op_dev(cheetah_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    CHECK_CTX(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!a || !out) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    hipLaunchKernelGGL(k_debug_op, dim3(lane_grid(ctx, n, 4)), dim3(BLOCK), 0, ctx->stream, op, a, b, out, n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_ctx_set_reserved_cus(cheetah_ctx* ctx, int cus) {
    CHECK_CTX(ctx);
    if (cus < 0 || cus >= ctx->num_cus) return fail(ctx, CHEETAH_ERR_INVALID, "reserved CUs must be in [0, number of CUs)");
    ctx->reserved_cus = cus;
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_multiply_units_per_launch(cheetah_ctx* ctx, size_t n, int vartime, int out_form, size_t* units) {
    CHECK_CTX(ctx);
    if (!units) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    const int waves = (vartime && out_form != CHEETAH_AFFINE) ? CH_VT_WAVES : CH_CT_WAVES;
    const size_t per_launch = (size_t)lane_grid(ctx, n ? n : 1, waves) * BLOCK * ROUNDS_PER_LAUNCH;
    *units = n < per_launch ? n : per_launch;
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_calibrate_mad_peak(cheetah_ctx* ctx, double* tera_mads_per_s) {
    CHECK_CTX(ctx);
    if (!tera_mads_per_s) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    u64* d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, 64));
    const int iters = 4096, blocks = ctx->num_cus * 8;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        hipLaunchKernelGGL(k_mad_peak, dim3(blocks), dim3(BLOCK), 0, ctx->stream, d, (u32)rep, iters);
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        if (rep > 0 && ms < best) best = ms;
    }
    HIP_TRY(ctx, hipFree(d));
    *tera_mads_per_s = (double)blocks * BLOCK * 64.0 * iters / (best * 1e-3) / 1e12;
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_timer_start(cheetah_ctx* ctx) {
    CHECK_CTX(ctx);
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_timer_stop_ms(cheetah_ctx* ctx, float* ms) {
    CHECK_CTX(ctx);
    if (!ms) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return CHEETAH_OK;
}
