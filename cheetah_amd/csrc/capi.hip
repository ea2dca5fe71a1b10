// C ABI of the batched Cheetah engine (include/cheetah_hip.h).  Host side only:
// argument checks, device buffers, copies, kernel launches.  No CPU fallback exists: a
// call either runs the HIP kernels or returns an error code.
//
// Every batch operation is described once (Op: its arrays, bytes per unit, and a launcher
// that enqueues the kernels for `cnt` units on one device context) and is reached three ways:
//   * `_dev` entry point, single-device context : the launcher, on the caller's device pointers
//   * `_dev` entry point, multi-device context  : run_dev_multi -- contiguous shards, peer copies
//   * host-pointer entry point                  : run_host -- per device a chunked pipeline
//     (upload chunk k + 1 / compute chunk k / download chunk k - 1 on three streams, staging
//     buffers kept in the context), one host thread per device when there are several
#include "../../include/cheetah_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

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

namespace {
struct DevSlot {  // grow-only device buffer owned by a context
    void* p = nullptr;
    size_t bytes = 0;
};
}  // namespace

struct cheetah_ctx {
    int device = -1;
    int shard_index = 0;  // 0 = the context the caller holds; i > 0 = its i-th further device
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    DeviceTables* tables = nullptr;
    u8* fixed_scalars = nullptr;  // q | cofactor (64 bytes), then generator | shift point (24 words)
    u64* scratch = nullptr;
    size_t scratch_words = 0;
    u64* noalloc_arena = nullptr;  // this device's address of noalloc_scratch (hipGetSymbolAddress, once, at creation)
    int num_cus = 0;
    bool force_reference_schedule = false;  // cheetah_debug_set_reference_schedule: no composed fast paths
    int reserved_cus = 0;  // CUs' worth of workgroup slots the persistent launches leave free (cheetah_ctx_set_reserved_cus)
    int launch_rounds = 2;  // rounds of the resident grid one ladder launch covers (cheetah_ctx_set_launch_rounds)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_order = nullptr;
    // host-pointer pipeline: copy streams, per-buffer-set events, staging buffers (two sets)
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    std::vector<DevSlot> stage[2];
    std::vector<DevSlot> pin[2];  // page-locked host staging of the host-pointer pipeline (hipHostMalloc), same indexing as stage
    // multi-device: the caller's context is shard 0 and owns one sub-context per further device
    std::vector<cheetah_ctx*> subs;
    DevSlot pair_tmp;           // multiply_double_with_basepoint_vartime, affine results: split scalars and the two partial points
    DevSlot redo_flags;         // vartime fast paths: one byte per wave of a launch, set for the waves left to the NAF kernels
    std::vector<DevSlot> peer;  // a sub-context's staging for `_dev` calls (its shard of the caller's device arrays)
    hipEvent_t ev_peer = nullptr;
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

static bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }
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

static u64* tab_field(const cheetah_ctx* c, size_t offset) {
    return reinterpret_cast<u64*>(reinterpret_cast<char*>(c->tables) + offset);
}
#define TAB(c, field) tab_field(c, offsetof(DeviceTables, field))

extern "C" const char* cheetah_version(void) { return "cheetah-hip 0.2 (gfx950)"; }

extern "C" const char* cheetah_last_error(const cheetah_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_create_err.c_str();
}

// ------------------------------------------------------------------ context -----

static void destroy_one(cheetah_ctx* ctx) {
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->s_in) (void)hipStreamSynchronize(ctx->s_in);
    if (ctx->s_out) (void)hipStreamSynchronize(ctx->s_out);
    if (ctx->tables) (void)hipFree(ctx->tables);
    if (ctx->fixed_scalars) (void)hipFree(ctx->fixed_scalars);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (int s = 0; s < 2; s++)
        for (auto& d : ctx->stage[s])
            if (d.p) (void)hipFree(d.p);
    for (int s = 0; s < 2; s++)
        for (auto& d : ctx->pin[s])
            if (d.p) (void)hipHostFree(d.p);
    for (auto& d : ctx->peer)
        if (d.p) (void)hipFree(d.p);
    if (ctx->pair_tmp.p) (void)hipFree(ctx->pair_tmp.p);
    if (ctx->redo_flags.p) (void)hipFree(ctx->redo_flags.p);
    hipEvent_t evs[] = {ctx->ev0, ctx->ev1, ctx->ev_order, ctx->ev_peer, ctx->ev_in[0], ctx->ev_in[1],
                        ctx->ev_k[0], ctx->ev_k[1], ctx->ev_out[0], ctx->ev_out[1]};
    for (hipEvent_t e : evs)
        if (e) (void)hipEventDestroy(e);
    if (ctx->s_in) (void)hipStreamDestroy(ctx->s_in);
    if (ctx->s_out) (void)hipStreamDestroy(ctx->s_out);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

static int32_t create_fail(cheetah_ctx* ctx, int32_t code, const std::string& msg) {
    {
        std::lock_guard<std::mutex> lk(g_err_mu);
        g_create_err = msg;
    }
    if (ctx) destroy_one(ctx);
    return code;
}

// One device: streams, events, the reference's lazy_static tables built on the device, the ladder kernels'
// window-table arena for the largest resident grid (so that no later `_dev` call has to allocate or synchronise).
static int32_t create_one(int device_id, int shard_index, cheetah_ctx** out) {
    cheetah_ctx* ctx = new (std::nothrow) cheetah_ctx();
    if (!ctx) return create_fail(nullptr, CHEETAH_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
    ctx->shard_index = shard_index;
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
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    CREATE_TRY(hipEventCreate(&ctx->ev0));
    CREATE_TRY(hipEventCreate(&ctx->ev1));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_peer, hipEventDisableTiming));
    for (int s = 0; s < 2; s++) {
        CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_in[s], hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_k[s], hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_out[s], hipEventDisableTiming));
    }
    CREATE_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&ctx->noalloc_arena), HIP_SYMBOL(noalloc_scratch)));
    CREATE_TRY(hipMalloc(&ctx->tables, sizeof(DeviceTables)));
    CREATE_TRY(hipMalloc(&ctx->fixed_scalars, 64 + 24 * sizeof(u64)));
    u64* seeds = reinterpret_cast<u64*>(ctx->fixed_scalars + 64);  // generator | shift point, read by the table kernels
    CREATE_TRY(hipMemcpyAsync(ctx->fixed_scalars, FQ_MODULUS_BYTES, 32, hipMemcpyHostToDevice, ctx->stream));
    CREATE_TRY(hipMemcpyAsync(ctx->fixed_scalars + 32, COFACTOR_BYTES, 32, hipMemcpyHostToDevice, ctx->stream));
    CREATE_TRY(hipMemcpyAsync(seeds, GENERATOR_XY, 12 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    CREATE_TRY(hipMemcpyAsync(seeds + 12, SHIFT_XY, 12 * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_build_tables, dim3(1), dim3(BUILD_TABLES_THREADS), 0, ctx->stream, ctx->tables, seeds, seeds + 12);
    hipLaunchKernelGGL((k_build_comb_table<0>), dim3(1), dim3(64), 0, ctx->stream, TAB(ctx, basepoint), seeds);
    hipLaunchKernelGGL((k_build_wide_table<0>), dim3(WIDE_WINDOWS), dim3(64), 0, ctx->stream, TAB(ctx, wide), seeds);
    CREATE_TRY(hipGetLastError());
    const int waves = CH_CT_WAVES > CH_VT_WAVES ? CH_CT_WAVES : CH_VT_WAVES;
    ctx->scratch_words = (size_t)ctx->num_cus * waves * LaneScratch::WORDS_PER_BLOCK;
    CREATE_TRY(hipMalloc(&ctx->scratch, ctx->scratch_words * sizeof(u64)));
    // one byte per wave of the largest launch (256 rounds of 1 024 workgroups): the vartime fast paths' verdicts; allocated
    // here so that no _dev call has to
    ctx->redo_flags.bytes = ((size_t)1 << 20) + 64;
    CREATE_TRY(hipMalloc(&ctx->redo_flags.p, ctx->redo_flags.bytes));
    CREATE_TRY(hipStreamSynchronize(ctx->stream));
#undef CREATE_TRY
    *out = ctx;
    return CHEETAH_OK;
}

static int32_t create_multi(const int* device_ids, int n_dev, bool allow_duplicates, cheetah_ctx** out) {
    if (!out) return create_fail(nullptr, CHEETAH_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!device_ids || n_dev < 1 || n_dev > 64) return create_fail(nullptr, CHEETAH_ERR_INVALID, "device_ids is NULL or n_dev is not in [1, 64]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return create_fail(nullptr, CHEETAH_ERR_NO_DEVICE, "no HIP device visible");
    for (int i = 0; i < n_dev; i++) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev)
            return create_fail(nullptr, CHEETAH_ERR_INVALID, "device id " + std::to_string(device_ids[i]) + " out of range (" + std::to_string(ndev) + " visible)");
        for (int j = 0; j < i && !allow_duplicates; j++)
            if (device_ids[j] == device_ids[i])
                return create_fail(nullptr, CHEETAH_ERR_INVALID, "device id " + std::to_string(device_ids[i]) + " listed twice");
    }
    cheetah_ctx* front = nullptr;
    if (int32_t rc = create_one(device_ids[0], 0, &front)) return rc;
    for (int i = 1; i < n_dev; i++) {
        cheetah_ctx* sub = nullptr;
        if (int32_t rc = create_one(device_ids[i], i, &sub)) {
            for (cheetah_ctx* s : front->subs) destroy_one(s);
            destroy_one(front);
            return rc;
        }
        front->subs.push_back(sub);
        if (device_ids[i] != device_ids[0]) {  // direct xGMI copies between the shards' devices where the platform allows
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[0]) == hipSuccess && can) {
                (void)hipSetDevice(device_ids[i]);
                (void)hipDeviceEnablePeerAccess(device_ids[0], 0);
                (void)hipSetDevice(device_ids[0]);
                (void)hipDeviceEnablePeerAccess(device_ids[i], 0);
                (void)hipGetLastError();  // "already enabled" is not an error here
            }
        }
    }
    (void)hipSetDevice(device_ids[0]);
    *out = front;
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_ctx_create(const int* device_ids, int n_dev, cheetah_ctx** out) {
    return create_multi(device_ids, n_dev, false, out);
}

extern "C" int32_t cheetah_debug_ctx_create_replicated(int device_id, int n_shards, cheetah_ctx** out) {
    if (n_shards < 1 || n_shards > 64) return create_fail(nullptr, CHEETAH_ERR_INVALID, "n_shards is not in [1, 64]");
    std::vector<int> ids((size_t)n_shards, device_id);
    return create_multi(ids.data(), n_shards, true, out);
}

extern "C" int32_t cheetah_ctx_destroy(cheetah_ctx* ctx) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    for (cheetah_ctx* s : ctx->subs) destroy_one(s);
    destroy_one(ctx);
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_ctx_num_devices(const cheetah_ctx* ctx) { return ctx ? 1 + (int32_t)ctx->subs.size() : CHEETAH_ERR_INVALID; }

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
    for (cheetah_ctx* s : ctx->subs) {
        HIP_TRY(ctx, hipSetDevice(s->device));
        HIP_TRY(ctx, hipStreamSynchronize(s->stream));
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_debug_set_reference_schedule(cheetah_ctx* ctx, int on) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    ctx->force_reference_schedule = on != 0;
    for (cheetah_ctx* s : ctx->subs) s->force_reference_schedule = on != 0;
    return CHEETAH_OK;
}

// The constant-time ladders choose, per wave, between two constructions of the window tables (shared-inversion affine
// steps; the reference's complete additions for a wave that holds the identity, a point of small order or one off the
// curve): timing then depends on the POINTS.  on != 0 removes the choice -- every wave takes the reference's
// construction (4 % slower on cheetah_multiply_batch: 29.2 against 30.3 M/s) -- for callers whose points are secret too.
extern "C" int32_t cheetah_ctx_set_point_independent_tables(cheetah_ctx* ctx, int on) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    const u64 flags = on ? 1 : 0;
    std::vector<cheetah_ctx*> all = {ctx};
    all.insert(all.end(), ctx->subs.begin(), ctx->subs.end());
    for (cheetah_ctx* c : all) {
        HIP_TRY(ctx, hipSetDevice(c->device));
        HIP_TRY(ctx, hipMemcpyAsync(&c->tables->flags, &flags, sizeof(flags), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(ctx, hipStreamSynchronize(c->stream));  // `flags` is a local
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_ctx_set_reserved_cus(cheetah_ctx* ctx, int cus) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (cus < 0 || cus >= ctx->num_cus) return fail(ctx, CHEETAH_ERR_INVALID, "reserved CUs must be in [0, number of CUs)");
    ctx->reserved_cus = cus;
    for (cheetah_ctx* s : ctx->subs) s->reserved_cus = cus;
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_ctx_set_launch_rounds(cheetah_ctx* ctx, int rounds) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (rounds < 1 || rounds > 256) return fail(ctx, CHEETAH_ERR_INVALID, "launch rounds must be in [1, 256]");
    ctx->launch_rounds = rounds;
    for (cheetah_ctx* s : ctx->subs) s->launch_rounds = rounds;
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

static int ew_grid(const cheetah_ctx* c, size_t n) {
    size_t tiles = (n + BLOCK - 1) / BLOCK, cap = (size_t)c->num_cus * 8;
    return (int)(tiles < cap ? tiles : cap);
}


// The arena is sized at creation for the single-scalar ladders; only the multi-scalar kernels with many points per
// group need more, and then only the first time: the context's stream is drained before the old arena is released.
static int32_t ensure_scratch(cheetah_ctx* ctx, size_t words) {
    if (ctx->scratch_words >= words) return CHEETAH_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) HIP_TRY(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_words = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->scratch, words * sizeof(u64)));
    ctx->scratch_words = words;
    return CHEETAH_OK;
}

static int32_t ensure_slot(cheetah_ctx* c, DevSlot& d, size_t bytes, hipStream_t user) {
    if (d.bytes >= bytes) return CHEETAH_OK;
    if (d.p) {
        HIP_TRY(c, hipStreamSynchronize(user));
        HIP_TRY(c, hipFree(d.p));
        d.p = nullptr;
        d.bytes = 0;
    }
    HIP_TRY(c, hipMalloc(&d.p, bytes));
    d.bytes = bytes;
    return CHEETAH_OK;
}

#define SET_DEVICE(ctx) HIP_TRY(ctx, hipSetDevice((ctx)->device))

// ------------------------------------------------------- generic dispatch -----

namespace {
struct Io {
    const void* in = nullptr;  // input array, or
    void* out = nullptr;       // output array
    size_t stride = 0;         // bytes per unit
    bool is_out = false;
    bool optional = false;     // output the caller may pass as NULL
    bool dev_always = false;   // host-pointer path: the kernels need a device array even when the caller's is NULL
};
Io in_(const void* p, size_t stride) {
    Io io;
    io.in = p;
    io.stride = stride;
    return io;
}
Io out_(void* p, size_t stride, bool optional = false, bool dev_always = false) {
    Io io;
    io.out = p;
    io.stride = stride;
    io.is_out = true;
    io.optional = optional;
    io.dev_always = dev_always;
    return io;
}
using LaunchFn = std::function<int32_t(cheetah_ctx*, void* const*, size_t)>;
struct Op {
    size_t n = 0;
    std::vector<Io> io;
    size_t chunk = (size_t)1 << 20;  // units per pipeline chunk of the host-pointer path (a multiple of 256)
    LaunchFn launch;                 // enqueue the kernels for `cnt` units on context c's stream; d[j] = device array j
};
constexpr size_t SHARD_ALIGN = 256;  // shard boundaries: every array stays 16-byte aligned at them
}  // namespace

static void shard_range(size_t n, size_t nsh, size_t i, size_t* off, size_t* cnt) {
    size_t per = (n + nsh - 1) / nsh;
    per = (per + SHARD_ALIGN - 1) / SHARD_ALIGN * SHARD_ALIGN;
    *off = i * per < n ? i * per : n;
    *cnt = n - *off < per ? n - *off : per;
}

static int32_t ensure_pinned(cheetah_ctx* c, DevSlot& d, size_t bytes) {
    if (d.bytes >= bytes) return CHEETAH_OK;
    if (d.p) {
        HIP_TRY(c, hipHostFree(d.p));
        d.p = nullptr;
        d.bytes = 0;
    }
    HIP_TRY(c, hipHostMalloc(&d.p, bytes, hipHostMallocDefault));
    d.bytes = bytes;
    return CHEETAH_OK;
}

// memcpy on up to four threads (the calling one included): one thread moves 10 - 20 GB/s, a PCIe 5 link 57.
static void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    constexpr size_t MIN_SLICE = (size_t)4 << 20;
    const unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw >= 4 || hw == 0 ? 4 : hw;
    if (bytes / MIN_SLICE < nt) nt = bytes / MIN_SLICE;
    if (nt < 2) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    size_t done = per;  // slice 0 is the caller's
    try {
        for (size_t i = 1; i < nt && done < bytes; i++, done += per) {
            const size_t o = done, m = bytes - o < per ? bytes - o : per;
            th.emplace_back([=]() { std::memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, m); });
        }
    } catch (...) {  // no thread to be had: the caller copies the rest
    }
    std::memcpy(dst, src, per < bytes ? per : bytes);
    if (done < bytes) std::memcpy(static_cast<char*>(dst) + done, static_cast<const char*>(src) + done, bytes - done);
    for (auto& t : th) t.join();
}

// Host-pointer path on ONE device: units [off, off + cnt) of the caller's host arrays.
// Host order per chunk: upload k + 1, launch k + 1, download k -- so that while the host is
// held by a download, the next chunk's kernels are already queued behind the running ones.
// Two staging sets on the device; events order reuse of a set.  Chunks of 8 MB and more go
// through page-locked staging on the host as well (PINNED): the caller's (pageable) arrays are
// copied to / from it by parallel_memcpy and the DMA engines move page-locked memory at the
// link's rate in both directions at once (57 GB/s each way on this box against 18 down / 31 up
// for the runtime's own pageable copies); smaller calls keep the direct copies and allocate
// nothing on the host.
static int32_t host_pipeline(cheetah_ctx* c, const Op& op, size_t off, size_t cnt) {
    SET_DEVICE(c);
    if (cnt == 0) return CHEETAH_OK;
    const size_t nio = op.io.size();
    auto used = [&](const Io& io) { return !io.is_out || io.out || io.dev_always; };
    auto moved = [&](const Io& io) { return !io.is_out || io.out; };  // crosses the link
    size_t unit_bytes = 0;
    for (const Io& io : op.io)
        if (moved(io)) unit_bytes += io.stride;
    size_t chunk = op.chunk < cnt ? op.chunk : cnt;
    while (chunk > 512 && chunk * unit_bytes > ((size_t)160 << 20)) chunk = (chunk / 2 + 255) / 256 * 256;  // bounds the staging
    const size_t nch = (cnt + chunk - 1) / chunk;
    const int sets = nch > 1 ? 2 : 1;
    bool pinned = chunk * unit_bytes >= ((size_t)8 << 20);
    for (int s = 0; s < sets; s++) {
        if (c->stage[s].size() < nio) c->stage[s].resize(nio);
        if (c->pin[s].size() < nio) c->pin[s].resize(nio);
        for (size_t j = 0; j < nio; j++) {
            if (used(op.io[j]))
                if (int32_t rc = ensure_slot(c, c->stage[s][j], chunk * op.io[j].stride, c->stream)) return rc;
            if (pinned && moved(op.io[j]) && ensure_pinned(c, c->pin[s][j], chunk * op.io[j].stride)) {
                // no page-locked memory to be had (a locked-memory limit, a busy host): the direct copies still work
                (void)hipGetLastError();
                c->err.clear();
                pinned = false;
            }
        }
    }
    std::vector<void*> d(nio);
    auto span = [&](size_t k, size_t* e0, size_t* m) {
        *e0 = off + k * chunk;
        *m = cnt - k * chunk < chunk ? cnt - k * chunk : chunk;
    };
    auto upload = [&](size_t k) -> int32_t {
        const int s = (int)(k & 1);
        size_t e0, m;
        span(k, &e0, &m);
        if (pinned && k >= 2) HIP_TRY(c, hipEventSynchronize(c->ev_in[s]));     // the set's previous upload has left the host staging
        if (k >= 2) HIP_TRY(c, hipStreamWaitEvent(c->s_in, c->ev_out[s], 0));  // the set's previous results have left the device staging
        for (size_t j = 0; j < nio; j++)
            if (!op.io[j].is_out) {
                const void* src = static_cast<const char*>(op.io[j].in) + e0 * op.io[j].stride;
                if (pinned) {
                    parallel_memcpy(c->pin[s][j].p, src, m * op.io[j].stride);
                    src = c->pin[s][j].p;
                }
                HIP_TRY(c, hipMemcpyAsync(c->stage[s][j].p, src, m * op.io[j].stride, hipMemcpyHostToDevice, c->s_in));
            }
        HIP_TRY(c, hipEventRecord(c->ev_in[s], c->s_in));
        return CHEETAH_OK;
    };
    auto compute = [&](size_t k) -> int32_t {
        const int s = (int)(k & 1);
        size_t e0, m;
        span(k, &e0, &m);
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_in[s], 0));
        if (k >= 2) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_out[s], 0));
        for (size_t j = 0; j < nio; j++) d[j] = used(op.io[j]) ? c->stage[s][j].p : nullptr;
        if (int32_t rc = op.launch(c, d.data(), m)) return rc;
        HIP_TRY(c, hipEventRecord(c->ev_k[s], c->stream));
        return CHEETAH_OK;
    };
    auto download = [&](size_t k) -> int32_t {  // enqueue: device staging -> the caller's array, or -> host staging
        const int s = (int)(k & 1);
        size_t e0, m;
        span(k, &e0, &m);
        HIP_TRY(c, hipStreamWaitEvent(c->s_out, c->ev_k[s], 0));
        for (size_t j = 0; j < nio; j++)
            if (op.io[j].is_out && op.io[j].out) {
                void* dst = pinned ? c->pin[s][j].p : static_cast<char*>(op.io[j].out) + e0 * op.io[j].stride;
                HIP_TRY(c, hipMemcpyAsync(dst, c->stage[s][j].p, m * op.io[j].stride, hipMemcpyDeviceToHost, c->s_out));
            }
        HIP_TRY(c, hipEventRecord(c->ev_out[s], c->s_out));
        return CHEETAH_OK;
    };
    auto deliver = [&](size_t k) -> int32_t {  // PINNED: host staging -> the caller's array, once the chunk has arrived
        const int s = (int)(k & 1);
        size_t e0, m;
        span(k, &e0, &m);
        HIP_TRY(c, hipEventSynchronize(c->ev_out[s]));
        for (size_t j = 0; j < nio; j++)
            if (op.io[j].is_out && op.io[j].out)
                parallel_memcpy(static_cast<char*>(op.io[j].out) + e0 * op.io[j].stride, c->pin[s][j].p, m * op.io[j].stride);
        return CHEETAH_OK;
    };
    // PINNED with several chunks: a second host thread delivers chunk k while this one stages chunk k + 2, so that
    // neither the kernels nor the DMA engines wait for a host copy (a trace of mul_by_generator at 2^22 showed the
    // device idle 0.7 ms in every 3.5 while one thread did both).
    struct Delivery {
        std::mutex mu;
        std::condition_variable cv;
        size_t enqueued = 0, delivered = 0;
        bool stop = false;
        int32_t rc = CHEETAH_OK;
        std::string err;
    } dl;
    const bool threaded = pinned && nch > 1;
    std::thread worker;
    if (threaded) {
        try {
            worker = std::thread([&]() {
                (void)hipSetDevice(c->device);
                for (size_t k = 0; k < nch; k++) {
                    {
                        std::unique_lock<std::mutex> lk(dl.mu);
                        dl.cv.wait(lk, [&] { return dl.enqueued > k || dl.stop; });
                        if (dl.enqueued <= k) return;
                    }
                    const int s = (int)(k & 1);
                    size_t e0, m;
                    span(k, &e0, &m);
                    const hipError_t e = hipEventSynchronize(c->ev_out[s]);
                    if (e == hipSuccess)
                        for (size_t j = 0; j < nio; j++)
                            if (op.io[j].is_out && op.io[j].out)
                                parallel_memcpy(static_cast<char*>(op.io[j].out) + e0 * op.io[j].stride, c->pin[s][j].p, m * op.io[j].stride);
                    {
                        std::lock_guard<std::mutex> lk(dl.mu);
                        if (e != hipSuccess) {
                            dl.rc = CHEETAH_ERR_HIP;
                            dl.err = std::string("hipEventSynchronize (delivery): ") + hipGetErrorString(e);
                            dl.delivered = nch;  // nothing more will be delivered: do not hold the main thread
                        } else {
                            dl.delivered = k + 1;
                        }
                    }
                    dl.cv.notify_all();
                    if (e != hipSuccess) return;
                }
            });
        } catch (...) {
            return fail(c, CHEETAH_ERR_NOMEM, "cannot start the delivery thread");
        }
    }
    int32_t rc = upload(0);
    if (!rc) rc = compute(0);
    for (size_t k = 0; k < nch && !rc; k++) {
        if (k + 1 < nch) {
            rc = upload(k + 1);
            if (!rc) rc = compute(k + 1);
        }
        if (!rc && threaded && k >= 2) {  // chunk k - 2 has left this set's host staging
            std::unique_lock<std::mutex> lk(dl.mu);
            dl.cv.wait(lk, [&] { return dl.delivered >= k - 1; });
            rc = dl.rc;
        }
        if (!rc) rc = download(k);
        if (!rc && threaded) {
            {
                std::lock_guard<std::mutex> lk(dl.mu);
                dl.enqueued = k + 1;
            }
            dl.cv.notify_all();
        } else if (!rc && pinned) {
            rc = deliver(k);
        }
    }
    if (threaded) {
        {
            std::lock_guard<std::mutex> lk(dl.mu);
            dl.stop = true;
        }
        dl.cv.notify_all();
        worker.join();
        if (!rc && dl.rc) {
            c->err = dl.err;
            rc = dl.rc;
        }
    }
    // never leave with copies in flight that touch the caller's memory, error or not
    hipError_t e1 = hipStreamSynchronize(c->s_in), e2 = hipStreamSynchronize(c->stream), e3 = hipStreamSynchronize(c->s_out);
    if (rc) return rc;
    HIP_TRY(c, e1);
    HIP_TRY(c, e2);
    HIP_TRY(c, e3);
    return CHEETAH_OK;
}

static int32_t run_host(cheetah_ctx* ctx, const Op& op) {
    const size_t nsh = 1 + ctx->subs.size();
    if (nsh == 1) return host_pipeline(ctx, op, 0, op.n);
    std::vector<int32_t> rcs(nsh, CHEETAH_OK);
    std::vector<std::thread> th;
    for (size_t i = 0; i < nsh; i++) {
        cheetah_ctx* c = i == 0 ? ctx : ctx->subs[i - 1];
        size_t off, cnt;
        shard_range(op.n, nsh, i, &off, &cnt);
        // a thread that cannot be started (std::system_error) must not unwind through the extern "C" caller with
        // earlier threads still joinable: that shard runs on the calling thread instead
        try {
            th.reserve(nsh);
            th.emplace_back([&rcs, &op, c, off, cnt, i]() { rcs[i] = host_pipeline(c, op, off, cnt); });
        } catch (...) {
            rcs[i] = host_pipeline(c, op, off, cnt);
        }
    }
    for (auto& t : th) t.join();
    (void)hipSetDevice(ctx->device);
    for (size_t i = 0; i < nsh; i++)
        if (rcs[i]) {
            if (i > 0) ctx->err = "device " + std::to_string(ctx->subs[i - 1]->device) + ": " + ctx->subs[i - 1]->err;
            return rcs[i];
        }
    return CHEETAH_OK;
}

// `_dev` entry point of a multi-device context: the arrays live on the first device.  Shard 0 runs in place;
// every further shard copies its slice to its own device (peer copy over xGMI), runs there, and copies its
// results back into the caller's arrays.  Asynchronous: the caller's stream waits for every shard on the device.
static int32_t run_dev_multi(cheetah_ctx* ctx, const Op& op) {
    const size_t nsh = 1 + ctx->subs.size(), nio = op.io.size();
    SET_DEVICE(ctx);
    HIP_TRY(ctx, hipEventRecord(ctx->ev_order, ctx->stream));  // the inputs are ready once the caller's stream gets here
    std::vector<void*> d(nio);
    for (size_t i = 1; i < nsh; i++) {
        cheetah_ctx* s = ctx->subs[i - 1];
        size_t off, cnt;
        shard_range(op.n, nsh, i, &off, &cnt);
        if (cnt == 0) continue;
        auto bail = [&](int32_t rc) {
            ctx->err = "device " + std::to_string(s->device) + ": " + s->err;
            (void)hipSetDevice(ctx->device);
            return rc;
        };
        if (hipSetDevice(s->device) != hipSuccess) return fail(ctx, CHEETAH_ERR_HIP, "hipSetDevice failed");
        if (s->peer.size() < nio) s->peer.resize(nio);
        for (size_t j = 0; j < nio; j++) {
            const Io& io = op.io[j];
            const bool used = !io.is_out || io.out;
            d[j] = nullptr;
            if (!used) continue;
            if (int32_t rc = ensure_slot(s, s->peer[j], cnt * io.stride, s->stream)) return bail(rc);
            d[j] = s->peer[j].p;
        }
        if (hipStreamWaitEvent(s->stream, ctx->ev_order, 0) != hipSuccess) return fail(ctx, CHEETAH_ERR_HIP, "hipStreamWaitEvent failed");
        for (size_t j = 0; j < nio; j++)
            if (!op.io[j].is_out)
                if (hipMemcpyPeerAsync(d[j], s->device, static_cast<const char*>(op.io[j].in) + off * op.io[j].stride, ctx->device,
                                       cnt * op.io[j].stride, s->stream) != hipSuccess)
                    return fail(ctx, CHEETAH_ERR_HIP, "hipMemcpyPeerAsync (inputs) failed");
        if (int32_t rc = op.launch(s, d.data(), cnt)) return bail(rc);
        for (size_t j = 0; j < nio; j++)
            if (op.io[j].is_out && op.io[j].out)
                if (hipMemcpyPeerAsync(static_cast<char*>(op.io[j].out) + off * op.io[j].stride, ctx->device, d[j], s->device,
                                       cnt * op.io[j].stride, s->stream) != hipSuccess)
                    return fail(ctx, CHEETAH_ERR_HIP, "hipMemcpyPeerAsync (results) failed");
        if (hipEventRecord(s->ev_peer, s->stream) != hipSuccess) return fail(ctx, CHEETAH_ERR_HIP, "hipEventRecord failed");
    }
    SET_DEVICE(ctx);
    size_t off0, cnt0;
    shard_range(op.n, nsh, 0, &off0, &cnt0);
    for (size_t j = 0; j < nio; j++) d[j] = op.io[j].is_out ? op.io[j].out : const_cast<void*>(op.io[j].in);
    if (cnt0)
        if (int32_t rc = op.launch(ctx, d.data(), cnt0)) return rc;
    for (size_t i = 1; i < nsh; i++) {
        size_t off, cnt;
        shard_range(op.n, nsh, i, &off, &cnt);
        if (cnt) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->subs[i - 1]->ev_peer, 0));
    }
    return CHEETAH_OK;
}

static int32_t run_op(cheetah_ctx* ctx, bool host, const Op& op) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (op.n == 0) return CHEETAH_OK;
    for (const Io& io : op.io)
        if ((!io.is_out && !io.in) || (io.is_out && !io.out && !io.optional)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (host) return run_host(ctx, op);
    if (!ctx->subs.empty()) return run_dev_multi(ctx, op);
    std::vector<void*> d(op.io.size());
    for (size_t j = 0; j < op.io.size(); j++) d[j] = op.io[j].is_out ? op.io[j].out : const_cast<void*>(op.io[j].in);
    return op.launch(ctx, d.data(), op.n);
}

template <class T>
static T* at(void* const* d, int j) {
    return static_cast<T*>(d[j]);
}

// ---------------------------------------------------------------- Fp6 -----

template <int OP>
static int32_t fp6_elementwise_impl(cheetah_ctx* ctx, const u64* a, const u64* b, u64* out, u8* is_some, size_t n) {
    SET_DEVICE(ctx);
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

static int32_t fp6_mul_op(cheetah_ctx* ctx, bool host, const u64* a, const u64* b, u64* out, size_t n) {
    Op op;
    op.n = n;
    op.io = {in_(a, 48), in_(b, 48), out_(out, 48)};
    op.launch = [](cheetah_ctx* c, void* const* d, size_t cnt) {
        return fp6_elementwise_impl<EW_MUL>(c, at<u64>(d, 0), at<u64>(d, 1), at<u64>(d, 2), nullptr, cnt);
    };
    return run_op(ctx, host, op);
}
template <int OP>
static int32_t fp6_unary_op(cheetah_ctx* ctx, bool host, const u64* a, u64* out, u8* is_some, size_t n) {
    Op op;
    op.n = n;
    op.io = {in_(a, 48), out_(out, 48)};
    if (OP != EW_SQUARE) op.io.push_back(out_(is_some, 1));
    op.launch = [](cheetah_ctx* c, void* const* d, size_t cnt) {
        return fp6_elementwise_impl<OP>(c, at<u64>(d, 0), nullptr, at<u64>(d, 1), OP != EW_SQUARE ? at<u8>(d, 2) : nullptr, cnt);
    };
    return run_op(ctx, host, op);
}

static int32_t fp_op_op(cheetah_ctx* ctx, bool host, int op, const u64* a, const u64* b, u64 scalar, u64* out, u8* is_some, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (op < 0 || op >= FP_OPS) return fail(ctx, CHEETAH_ERR_INVALID, "unknown cheetah_fp_op");
    const bool binary = op == FP_ADD || op == FP_SUB || op == FP_MUL;
    if (binary && !b && n) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    Op o;
    o.n = n;
    o.io = {in_(a, 8), in_(binary ? b : a, 8), out_(out, 8), out_(is_some, 1, /*optional*/ true)};
    o.launch = [op, scalar](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned8(d[0]) || !aligned8(d[1]) || !aligned8(d[2])) return fail(c, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
        hipLaunchKernelGGL(k_fp_op, dim3(ew_grid(c, cnt)), dim3(BLOCK), 0, c->stream, op, at<u64>(d, 0), at<u64>(d, 1), scalar, at<u64>(d, 2),
                           at<u8>(d, 3), cnt);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, o);
}
extern "C" int32_t cheetah_fp_op_batch_dev(cheetah_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t scalar, uint64_t* out,
                                           uint8_t* is_some, size_t n) {
    return fp_op_op(ctx, false, op, a, b, scalar, out, is_some, n);
}
extern "C" int32_t cheetah_fp_op_batch(cheetah_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t scalar, uint64_t* out,
                                       uint8_t* is_some, size_t n) {
    return fp_op_op(ctx, true, op, a, b, scalar, out, is_some, n);
}
static int32_t fp6_op_op(cheetah_ctx* ctx, bool host, int op, const u64* a, const u64* b, u64 scalar, u64* out, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (op < 0 || op >= F6_OPS) return fail(ctx, CHEETAH_ERR_INVALID, "unknown cheetah_fp6_op");
    const bool binary = op == F6_ADD || op == F6_SUB;
    if (binary && !b && n) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    Op o;
    o.n = n;
    if (binary) o.io = {in_(a, 48), in_(b, 48), out_(out, 48)};
    else o.io = {in_(a, 48), out_(out, 48)};
    o.launch = [op, binary, scalar](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        const u64* bb = binary ? at<u64>(d, 1) : nullptr;
        u64* oo = at<u64>(d, binary ? 2 : 1);
        if (!aligned16(d[0]) || !aligned16(oo) || (bb && !aligned16(bb))) return fail(c, CHEETAH_ERR_INVALID, "Fp6 device arrays must be 16-byte aligned");
        const size_t tiles = (cnt + BLOCK - 1) / BLOCK;
        hipLaunchKernelGGL(k_fp6_op, dim3((unsigned)(tiles < ((size_t)1 << 22) ? tiles : ((size_t)1 << 22))), dim3(BLOCK), 0, c->stream, op,
                           at<u64>(d, 0), bb, scalar, oo, cnt);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, o);
}
extern "C" int32_t cheetah_fp6_op_batch_dev(cheetah_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t scalar, uint64_t* out,
                                            size_t n) {
    return fp6_op_op(ctx, false, op, a, b, scalar, out, n);
}
extern "C" int32_t cheetah_fp6_op_batch(cheetah_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t scalar, uint64_t* out, size_t n) {
    return fp6_op_op(ctx, true, op, a, b, scalar, out, n);
}
// Fp6::exp / exp_vartime (src/fp6.rs:1053-1086) over a batch: one 384-bit exponent (HOST pointer, 6 limbs) for all elements.
static int32_t fp6_exp_op(cheetah_ctx* ctx, bool host, const u64* a, const u64* power6, u64* out, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (!power6) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    u64 pw[6];
    for (int i = 0; i < 6; i++) pw[i] = power6[i];
    Op o;
    o.n = n;
    o.chunk = (size_t)1 << 18;
    o.io = {in_(a, 48), out_(out, 48)};
    o.launch = [pw](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned16(d[0]) || !aligned16(d[1])) return fail(c, CHEETAH_ERR_INVALID, "Fp6 device arrays must be 16-byte aligned");
        hipLaunchKernelGGL(k_fp6_exp, dim3(ew_grid(c, cnt)), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), pw[0], pw[1], pw[2], pw[3], pw[4], pw[5],
                           at<u64>(d, 1), cnt);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, o);
}
extern "C" int32_t cheetah_fp6_exp_batch(cheetah_ctx* ctx, const uint64_t* a, const uint64_t* power6, uint64_t* out, size_t n) {
    return fp6_exp_op(ctx, true, a, power6, out, n);
}
extern "C" int32_t cheetah_fp6_exp_batch_dev(cheetah_ctx* ctx, const uint64_t* a, const uint64_t* power6, uint64_t* out, size_t n) {
    return fp6_exp_op(ctx, false, a, power6, out, n);
}
extern "C" int32_t cheetah_fp6_mul_batch_dev(cheetah_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    return fp6_mul_op(ctx, false, a, b, out, n);
}
extern "C" int32_t cheetah_fp6_mul_batch(cheetah_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    return fp6_mul_op(ctx, true, a, b, out, n);
}
extern "C" int32_t cheetah_fp6_square_batch_dev(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return fp6_unary_op<EW_SQUARE>(ctx, false, a, out, nullptr, n);
}
extern "C" int32_t cheetah_fp6_square_batch(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return fp6_unary_op<EW_SQUARE>(ctx, true, a, out, nullptr, n);
}
extern "C" int32_t cheetah_fp6_invert_batch_dev(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n) {
    return fp6_unary_op<EW_INVERT>(ctx, false, a, out, is_some, n);
}
extern "C" int32_t cheetah_fp6_invert_batch(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n) {
    return fp6_unary_op<EW_INVERT>(ctx, true, a, out, is_some, n);
}
extern "C" int32_t cheetah_fp6_sqrt_batch_dev(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n) {
    return fp6_unary_op<EW_SQRT>(ctx, false, a, out, is_some, n);
}
extern "C" int32_t cheetah_fp6_sqrt_batch(cheetah_ctx* ctx, const uint64_t* a, uint64_t* out, uint8_t* is_some, size_t n) {
    return fp6_unary_op<EW_SQRT>(ctx, true, a, out, is_some, n);
}

// ------------------------------------------------------ batch_normalize -----

template <int JAC>
static int32_t batch_normalize_impl(cheetah_ctx* ctx, const u64* xyz, u64* xy, u8* inf, size_t n) {
    SET_DEVICE(ctx);
    if (!aligned16(xyz) || !aligned16(xy)) return fail(ctx, CHEETAH_ERR_INVALID, "point arrays must be 16-byte aligned");
    constexpr int CHUNK = CH_BN_CHUNK;
    size_t nchunks = (n + CHUNK - 1) / CHUNK;
    int grid = lane_grid(ctx, nchunks, CH_BN_WAVES);
    hipLaunchKernelGGL((k_batch_normalize<JAC>), dim3(grid), dim3(BLOCK), 0, ctx->stream, xyz, xy, inf, n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}
template <int JAC>
static int32_t batch_normalize_op(cheetah_ctx* ctx, bool host, const u64* xyz, u64* xy, u8* inf, size_t n) {
    Op op;
    op.n = n;
    op.io = {in_(xyz, 144), out_(xy, 96), out_(inf, 1)};
    op.launch = [](cheetah_ctx* c, void* const* d, size_t cnt) {
        return batch_normalize_impl<JAC>(c, at<u64>(d, 0), at<u64>(d, 1), at<u8>(d, 2), cnt);
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_batch_normalize_projective_dev(cheetah_ctx* ctx, const uint64_t* xyz, uint64_t* xy, uint8_t* inf, size_t n) {
    return batch_normalize_op<0>(ctx, false, xyz, xy, inf, n);
}
extern "C" int32_t cheetah_batch_normalize_jacobian_dev(cheetah_ctx* ctx, const uint64_t* xyz, uint64_t* xy, uint8_t* inf, size_t n) {
    return batch_normalize_op<1>(ctx, false, xyz, xy, inf, n);
}
extern "C" int32_t cheetah_batch_normalize_projective(cheetah_ctx* ctx, const uint64_t* xyz, uint64_t* xy, uint8_t* inf, size_t n) {
    return batch_normalize_op<0>(ctx, true, xyz, xy, inf, n);
}
extern "C" int32_t cheetah_batch_normalize_jacobian(cheetah_ctx* ctx, const uint64_t* xyz, uint64_t* xy, uint8_t* inf, size_t n) {
    return batch_normalize_op<1>(ctx, true, xyz, xy, inf, n);
}

// ------------------------------------------------------------- ladders -----

// Which reference ladder a (in_form, out_form) pair names (include/cheetah_hip.h): the point a call returns is
// always the reference method's; with AFFINE results it does not depend on the accumulator's coordinates, and
// the modified-Jacobian ladder (JacobianPoint's, which AffinePoint's methods delegate to) is run because it is
// cheaper.  PROJECTIVE <-> JACOBIAN has no counterpart in the reference and is rejected.
static int32_t check_forms(cheetah_ctx* ctx, int in_form, int out_form) {
    if (in_form < CHEETAH_AFFINE || in_form > CHEETAH_JACOBIAN) return fail(ctx, CHEETAH_ERR_INVALID, "in_form must be AFFINE, PROJECTIVE or JACOBIAN");
    if (out_form < CHEETAH_AFFINE || out_form > CHEETAH_JACOBIAN) return fail(ctx, CHEETAH_ERR_INVALID, "out_form must be AFFINE, PROJECTIVE or JACOBIAN");
    if ((in_form == CHEETAH_PROJECTIVE && out_form == CHEETAH_JACOBIAN) || (in_form == CHEETAH_JACOBIAN && out_form == CHEETAH_PROJECTIVE))
        return fail(ctx, CHEETAH_ERR_INVALID, "PROJECTIVE and JACOBIAN cannot be mixed in one call");
    return CHEETAH_OK;
}
static size_t form_limbs(int form) { return form == CHEETAH_AFFINE ? 12 : 18; }

// Fixed base.  comb / wide: the base's BasePointTable (32 x LookupTable<8>) and its widened table; GENERIC: a
// caller-supplied base (tables may hold identities).
template <bool GENERIC>
static int32_t fixed_base_impl(cheetah_ctx* ctx, const u64* comb, const u64* wide, const u8* scalars32, u64* out, u8* inf, size_t n,
                               int out_form, bool reference_comb = false) {
    SET_DEVICE(ctx);
    if (!aligned8(scalars32) || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
    if (out_form != CHEETAH_JACOBIAN && !inf)
        return fail(ctx, CHEETAH_ERR_INVALID, "inf must not be NULL for BasePointTable forms: it carries the out-of-range flag 0xFF");
    if (out_form == CHEETAH_AFFINE && !(GENERIC && reference_comb)) {  // affine results: the widened comb (one 1 024-thread workgroup per CU)
        size_t want = (n + WIDE_BLOCK - 1) / WIDE_BLOCK;
        const size_t cus_w = (size_t)(ctx->num_cus - ctx->reserved_cus);
        int grid_w = (int)(want < cus_w ? want : cus_w);
        const size_t per_launch_w = (size_t)grid_w * WIDE_BLOCK * ctx->launch_rounds * 4;
        for (size_t off = 0; off < n; off += per_launch_w) {
            const size_t cnt = n - off < per_launch_w ? n - off : per_launch_w;
            hipLaunchKernelGGL((k_mul_by_generator_wide<GENERIC>), dim3(grid_w), dim3(WIDE_BLOCK), 0, ctx->stream, ctx->tables, wide,
                               scalars32 + off * 32, out + off * 12, inf + off, cnt);
        }
        HIP_TRY(ctx, hipGetLastError());
        return CHEETAH_OK;
    }
    int grid = lane_grid(ctx, n, CH_GEN_WAVES);
    const size_t per_launch = (size_t)grid * BLOCK * ctx->launch_rounds * 4;  // bounded launches, as in multiply_impl
    for (size_t off = 0; off < n; off += per_launch) {
        const size_t cnt = n - off < per_launch ? n - off : per_launch;
        const u8* sc_o = scalars32 + off * 32;
        u64* out_o = out + off * (out_form == CHEETAH_AFFINE ? 12 : 18);
        u8* inf_o = inf ? inf + off : nullptr;
        if constexpr (GENERIC) {
            if (out_form == CHEETAH_AFFINE) {  // an off-curve base: the reference's comb, normalised
                hipLaunchKernelGGL((k_mul_by_generator<0, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, comb, sc_o, out_o, inf_o, cnt);
                continue;
            }
        }
        if (out_form == CHEETAH_PROJECTIVE)
            hipLaunchKernelGGL((k_mul_by_generator<1, GENERIC>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, comb, sc_o, out_o, inf_o, cnt);
        else if (!GENERIC)
            hipLaunchKernelGGL((k_mul_by_generator_lookup<2>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, sc_o, out_o, inf_o, cnt);
    }
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

// BASEPOINT_TABLE's range check of the host-pointer entry points
static int32_t check_comb_range(cheetah_ctx* ctx, const u8* scalars32, size_t n) {
    for (size_t i = 0; i < n; i++)
        if ((scalars32[32 * i + 31] >> 4) > 7)
            return fail(ctx, CHEETAH_ERR_SCALAR, "scalar with top nibble > 7: outside BasePointTable's LookupTable<8> range");
    return CHEETAH_OK;
}

static int32_t mul_by_generator_op(cheetah_ctx* ctx, bool host, const u8* scalars32, u64* out, u8* inf, size_t n, int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, CHEETAH_AFFINE, out_form)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (host && out_form != CHEETAH_JACOBIAN && scalars32)  // BASEPOINT_LOOKUP's LookupTable<128> covers every radix-256 digit
        if (int32_t rc = check_comb_range(ctx, scalars32, n)) return rc;
    Op op;
    op.n = n;
    op.chunk = (size_t)1 << 18;
    op.io = {in_(scalars32, 32), out_(out, form_limbs(out_form) * 8),
             out_(inf, 1, /*optional*/ (host && out_form != CHEETAH_AFFINE) || out_form == CHEETAH_JACOBIAN,
                  /*dev_always*/ out_form != CHEETAH_JACOBIAN)};
    op.launch = [out_form](cheetah_ctx* c, void* const* d, size_t cnt) {
        return fixed_base_impl<false>(c, TAB(c, basepoint), TAB(c, wide), at<u8>(d, 0), at<u64>(d, 1), at<u8>(d, 2), cnt, out_form);
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_mul_by_generator_batch_dev(cheetah_ctx* ctx, const uint8_t* scalars32, uint64_t* out, uint8_t* inf,
                                                      size_t n, int out_form) {
    return mul_by_generator_op(ctx, false, scalars32, out, inf, n, out_form);
}
extern "C" int32_t cheetah_mul_by_generator_batch(cheetah_ctx* ctx, const uint8_t* scalars32, uint64_t* out, uint8_t* inf, size_t n,
                                                  int out_form) {
    return mul_by_generator_op(ctx, true, scalars32, out, inf, n, out_form);
}

// ---- BasePointTable::create(P) / multiply for a caller-supplied base (src/lookup.rs:164-206) ----

struct cheetah_basepoint_table {
    cheetah_ctx* ctx = nullptr;
    std::vector<u64*> comb, wide;  // one copy per shard of the context
    // A base that is NOT on the curve (an unchecked decoding can produce one), or one of small order / whose reference
    // table met the complete addition's exceptional pair (quirk Q7: an identity or a y = 0 among the comb's entries): the
    // reference's comb still returns a definite value for it, but the widened comb -- another schedule over other
    // multiples -- returns another, so AFFINE results then come from the reference's comb and a normalisation
    // (k_mul_by_generator<0, GENERIC>).  false in either case.
    bool on_curve = true;
};
constexpr size_t COMB_WORDS = 32 * 8 * 12, WIDE_WORDS = (size_t)WIDE_WINDOWS * WIDE_ENTRIES * 12;

extern "C" int32_t cheetah_basepoint_table_destroy(cheetah_basepoint_table* t) {
    if (!t) return CHEETAH_ERR_INVALID;
    for (size_t i = 0; i < t->comb.size(); i++) {
        cheetah_ctx* c = i == 0 ? t->ctx : t->ctx->subs[i - 1];
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (t->comb[i]) (void)hipFree(t->comb[i]);  // comb and wide are one allocation
    }
    (void)hipSetDevice(t->ctx->device);
    delete t;
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_basepoint_table_create(cheetah_ctx* ctx, const uint64_t* point, int in_form, cheetah_basepoint_table** out) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (!out || !point) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    *out = nullptr;
    if (in_form != CHEETAH_AFFINE && in_form != CHEETAH_PROJECTIVE)
        return fail(ctx, CHEETAH_ERR_INVALID, "BasePointTable is created from an AffinePoint or a ProjectivePoint (src/lookup.rs:138-173)");
    cheetah_basepoint_table* t = new (std::nothrow) cheetah_basepoint_table();
    if (!t) return fail(ctx, CHEETAH_ERR_NOMEM, "out of host memory");
    t->ctx = ctx;
    const size_t nsh = 1 + ctx->subs.size(), pt_words = form_limbs(in_form);
    t->comb.assign(nsh, nullptr);
    t->wide.assign(nsh, nullptr);
    for (size_t i = 0; i < nsh; i++) {
        cheetah_ctx* c = i == 0 ? ctx : ctx->subs[i - 1];
        u64* buf = nullptr;
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = hipMalloc(&buf, (COMB_WORDS + WIDE_WORDS + 24) * sizeof(u64));
        if (e == hipSuccess) {
            t->comb[i] = buf;
            t->wide[i] = buf + COMB_WORDS;
            u64* pt = buf + COMB_WORDS + WIDE_WORDS;
            e = hipMemcpyAsync(pt, point, pt_words * sizeof(u64), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) {
                if (in_form == CHEETAH_AFFINE) {
                    hipLaunchKernelGGL((k_build_comb_table<0>), dim3(1), dim3(64), 0, c->stream, t->comb[i], pt);
                    hipLaunchKernelGGL((k_build_wide_table<0>), dim3(WIDE_WINDOWS), dim3(64), 0, c->stream, t->wide[i], pt);
                } else {
                    hipLaunchKernelGGL((k_build_comb_table<1>), dim3(1), dim3(64), 0, c->stream, t->comb[i], pt);
                    hipLaunchKernelGGL((k_build_wide_table<1>), dim3(WIDE_WINDOWS), dim3(64), 0, c->stream, t->wide[i], pt);
                }
                u8* flag = reinterpret_cast<u8*>(pt + 20);
                if (in_form == CHEETAH_AFFINE) hipLaunchKernelGGL((k_is_on_curve<0>), dim3(1), dim3(BLOCK), 0, c->stream, pt, flag, (size_t)1);
                else hipLaunchKernelGGL((k_is_on_curve<1>), dim3(1), dim3(BLOCK), 0, c->stream, pt, flag, (size_t)1);
                hipLaunchKernelGGL(k_comb_table_is_ordinary, dim3(1), dim3(256), 0, c->stream, t->comb[i], flag + 1);
                e = hipGetLastError();
                u8 on[2] = {1, 1};  // on the curve; no identity and no y = 0 among the comb's 256 entries
                if (e == hipSuccess) e = hipMemcpyAsync(on, flag, 2, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e == hipSuccess && !(on[0] && on[1])) t->on_curve = false;
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // `point` is the caller's: done with it on return
        }
        if (e != hipSuccess) {
            ctx->err = std::string("cheetah_basepoint_table_create: ") + hipGetErrorString(e);
            cheetah_basepoint_table_destroy(t);
            return CHEETAH_ERR_HIP;
        }
    }
    (void)hipSetDevice(ctx->device);
    *out = t;
    return CHEETAH_OK;
}

static int32_t table_multiply_op(cheetah_ctx* ctx, bool host, const cheetah_basepoint_table* t, const u8* scalars32, u64* out, u8* inf,
                                 size_t n, int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (!t || t->ctx != ctx) return fail(ctx, CHEETAH_ERR_INVALID, "the table does not belong to this context");
    if (out_form != CHEETAH_AFFINE && out_form != CHEETAH_PROJECTIVE)
        return fail(ctx, CHEETAH_ERR_INVALID, "BasePointTable::multiply returns a ProjectivePoint (or its affine point): out_form must be AFFINE or PROJECTIVE");
    if (n == 0) return CHEETAH_OK;
    if (host && scalars32)
        if (int32_t rc = check_comb_range(ctx, scalars32, n)) return rc;
    Op op;
    op.n = n;
    op.chunk = (size_t)1 << 18;
    op.io = {in_(scalars32, 32), out_(out, form_limbs(out_form) * 8), out_(inf, 1, host && out_form != CHEETAH_AFFINE, true)};
    op.launch = [out_form, t](cheetah_ctx* c, void* const* d, size_t cnt) {
        return fixed_base_impl<true>(c, t->comb[c->shard_index], t->wide[c->shard_index], at<u8>(d, 0), at<u64>(d, 1), at<u8>(d, 2), cnt,
                                     out_form, !t->on_curve);
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_basepoint_table_multiply_batch(cheetah_ctx* ctx, const cheetah_basepoint_table* t, const uint8_t* scalars32,
                                                          uint64_t* out, uint8_t* inf, size_t n, int out_form) {
    return table_multiply_op(ctx, true, t, scalars32, out, inf, n, out_form);
}
extern "C" int32_t cheetah_basepoint_table_multiply_batch_dev(cheetah_ctx* ctx, const cheetah_basepoint_table* t,
                                                              const uint8_t* scalars32, uint64_t* out, uint8_t* inf, size_t n,
                                                              int out_form) {
    return table_multiply_op(ctx, false, t, scalars32, out, inf, n, out_form);
}
extern "C" int32_t cheetah_basepoint_table_read(cheetah_ctx* ctx, const cheetah_basepoint_table* t, uint64_t* out_xy, size_t max_points) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (!t || t->ctx != ctx || !out_xy) return fail(ctx, CHEETAH_ERR_INVALID, "bad table or null pointer");
    SET_DEVICE(ctx);
    const size_t pts = max_points < 256 ? max_points : 256;
    HIP_TRY(ctx, hipMemcpyAsync(out_xy, t->comb[0], pts * 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CHEETAH_OK;
}

// ---- variable base ----

// arena == nullptr: the context's own scratch; otherwise a caller-chosen arena that holds
// max_blocks workgroups' tables, and the grids are capped to it (the kernels grid-stride over the rest).
constexpr size_t POOL_ROUNDS_FACTOR = 8;
template <bool VARTIME>
static int32_t multiply_impl(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* scalars32, u64* out, u8* inf, size_t n,
                             int out_form, u64* arena = nullptr, int max_blocks = 0) {
    SET_DEVICE(ctx);
    if (out_form == CHEETAH_AFFINE && !inf) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned8(pts) || !aligned8(scalars32) || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
    // multiply_vartime with AFFINE results: the fixed-window ladder with indexed lookups (k_multiply_ct<..., DIRECT>)
    // for every wave whose scalars are all below 2^255, then the reference's NAF ladder for the remaining waves.
    const bool direct = VARTIME && out_form == CHEETAH_AFFINE;
    // multiply_vartime with the reference's own representative (L2): the pooled NAF kernel, VtPool::SLOTS x 64 items per wave
    const bool pooled = VARTIME && !direct;
    const size_t pool_words = VtPool::words(out_form == CHEETAH_PROJECTIVE ? 3 : 4);
    // the ONLY_WIDE instance of the NAF kernel (normally nothing to do) is compiled for 2 waves
    int grid_vt = pooled ? lane_grid(ctx, (n + VtPool::SLOTS - 1) / VtPool::SLOTS, CH_VT_POOL_WAVES) : lane_grid(ctx, n, direct ? 2 : CH_VT_WAVES);
    int grid_ct = lane_grid(ctx, n, CH_CT_WAVES);
    if (arena) {
        if (grid_vt > max_blocks) grid_vt = max_blocks;
        if (grid_ct > max_blocks) grid_ct = max_blocks;
    } else {
        size_t words = (size_t)(grid_vt > grid_ct ? grid_vt : grid_ct) * LaneScratch::WORDS_PER_BLOCK;
        if (pooled) words = (size_t)grid_vt * pool_words;
        if (int32_t rc = ensure_scratch(ctx, words)) return rc;
        arena = ctx->scratch;
    }
    const int grid = pooled ? grid_vt : grid_ct;
    // One launch covers at most ctx->launch_rounds rounds of the resident grid (default 2: 17 ms of work at 4 workgroups per CU).
    // The workgroups are persistent within a launch and fill every CU, so a kernel on another stream (the RCCL
    // all-gather of the previous chunk) can only get CU slots at a launch boundary: bounding the launch bounds its wait.
    const size_t in_w = form_limbs(in_form), out_w = form_limbs(out_form);
    // The pooled kernel's rounds end wave by wave (each wave's NAFs have their own length), so a launch boundary costs it
    // more than it costs the lock-step ladders: measured at 2^22, projective representative, 20.6 M/s at 2 rounds per launch,
    // 21.7 at 16 (profiles/r03_ladder_timings.txt).  Its launches therefore cover POOL_ROUNDS_FACTOR times as many rounds.
    const size_t per_launch = (size_t)grid * BLOCK * ctx->launch_rounds * (pooled ? VtPool::SLOTS * POOL_ROUNDS_FACTOR : 1);
    u8* flags = nullptr;  // the fast path's verdict per wave of a launch, read by the NAF kernel that follows it
    if (direct) {
        if (int32_t rc = ensure_slot(ctx, ctx->redo_flags, per_launch / 64 + 1, ctx->stream)) return rc;
        flags = static_cast<u8*>(ctx->redo_flags.p);
    }
    for (size_t off = 0; off < n; off += per_launch) {
        const size_t cnt = n - off < per_launch ? n - off : per_launch;
        const u64* pts_o = pts + off * in_w;
        const u8* sc_o = scalars32 + off * 32;
        u64* out_o = out + off * out_w;
        u8* inf_o = inf ? inf + off : nullptr;
#define LAUNCH(A, I, O)                                                                                                        \
    do {                                                                                                                       \
        if constexpr (VARTIME && O == 0) {                                                                                     \
            hipLaunchKernelGGL((k_multiply_ct<A, I, O, true>), dim3(grid_ct), dim3(BLOCK), 0, ctx->stream, ctx->tables, pts_o,  \
                               sc_o, out_o, inf_o, cnt, arena, flags);                                                         \
            hipLaunchKernelGGL((k_multiply_vartime<A, I, O, true>), dim3(grid_vt), dim3(BLOCK), 0, ctx->stream, ctx->tables,    \
                               pts_o, sc_o, out_o, inf_o, cnt, arena, 0, flags);                                               \
        } else if constexpr (VARTIME)                                                                                          \
            hipLaunchKernelGGL((k_multiply_vartime_pool<A, I, O>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables,        \
                               pts_o, sc_o, out_o, inf_o, cnt, arena, pool_words);                                             \
        else                                                                                                                   \
            hipLaunchKernelGGL((k_multiply_ct<A, I, O, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables, pts_o,    \
                               sc_o, out_o, inf_o, cnt, arena);                                                                \
    } while (0)
        switch (in_form * 3 + out_form) {
            case 0: LAUNCH(MJac, 0, 0); break;  /* A -> A */
            case 1: LAUNCH(Proj, 0, 1); break;  /* A -> P */
            case 2: LAUNCH(MJac, 0, 2); break;  /* A -> J */
            case 3: LAUNCH(MJac, 1, 0); break;  /* P -> A: the affine point is the same on either accumulator; MJac is cheaper */
            case 4: LAUNCH(Proj, 1, 1); break;  /* P -> P */
            case 6: LAUNCH(MJac, 2, 0); break;  /* J -> A */
            default: LAUNCH(MJac, 2, 2); break; /* J -> J */
        }
#undef LAUNCH
    }
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

template <bool VARTIME>
static int32_t multiply_op(cheetah_ctx* ctx, bool host, const u64* pts, int in_form, const u8* scalars32, u64* out, u8* inf, size_t n,
                           int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    Op op;
    op.n = n;
    op.chunk = (size_t)1 << 18;
    op.io = {in_(pts, form_limbs(in_form) * 8), in_(scalars32, 32), out_(out, form_limbs(out_form) * 8),
             out_(inf, 1, out_form != CHEETAH_AFFINE)};
    op.launch = [in_form, out_form](cheetah_ctx* c, void* const* d, size_t cnt) {
        return multiply_impl<VARTIME>(c, at<u64>(d, 0), in_form, at<u8>(d, 1), at<u64>(d, 2), at<u8>(d, 3), cnt, out_form);
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_multiply_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32,
                                              uint64_t* out, uint8_t* inf, size_t n, int out_form) {
    return multiply_op<false>(ctx, false, pts, in_form, scalars32, out, inf, n, out_form);
}
extern "C" int32_t cheetah_multiply_vartime_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form,
                                                      const uint8_t* scalars32, uint64_t* out, uint8_t* inf, size_t n,
                                                      int out_form) {
    return multiply_op<true>(ctx, false, pts, in_form, scalars32, out, inf, n, out_form);
}
extern "C" int32_t cheetah_multiply_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32, uint64_t* out,
                                          uint8_t* inf, size_t n, int out_form) {
    return multiply_op<false>(ctx, true, pts, in_form, scalars32, out, inf, n, out_form);
}
extern "C" int32_t cheetah_multiply_vartime_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32,
                                                  uint64_t* out, uint8_t* inf, size_t n, int out_form) {
    return multiply_op<true>(ctx, true, pts, in_form, scalars32, out, inf, n, out_form);
}

// cheetah_multiply_batch_dev for callers that cannot allocate: the same kernels on the file-scope arena.
// Calls no hipMalloc / hipFree (nor anything else that allocates): argument checks, hipSetDevice, kernel launches.
// The arena needs no clearing: a workgroup writes every table entry it later reads, in every round.
// Runs on the context's first device only (a multi-device context's other shards are not used).
extern "C" int32_t cheetah_multiply_batch_dev_noalloc(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32,
                                                      uint64_t* out, uint8_t* inf, size_t n, int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (n > (size_t)CHEETAH_NOALLOC_MAX_N)
        return fail(ctx, CHEETAH_ERR_INVALID, "cheetah_multiply_batch_dev_noalloc: n exceeds CHEETAH_NOALLOC_MAX_N (use cheetah_multiply_batch_dev)");
    if (!ctx->noalloc_arena) return fail(ctx, CHEETAH_ERR_HIP, "the static arena's device address is not known");
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    if (n == 0) return CHEETAH_OK;
    if (!pts || !scalars32 || !out) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    return multiply_impl<false>(ctx, pts, in_form, scalars32, out, inf, n, out_form, ctx->noalloc_arena, NOALLOC_MAX_BLOCKS);
}

// --------------------------------------------- multi-scalar ladders, checks -----

// multiply_double_with_basepoint_vartime with AFFINE results: out[i] = a_i P_i + b_i G as the sum of the two
// single-scalar fast paths (fixed-window ladder with indexed lookups for a P, widened comb for b G) instead of the
// reference's Straus NAF ladder: the affine point is the same for every pair of scalars below 2^255, and the waves
// that hold a larger one are redone by the NAF kernel afterwards.  Sub-batches of 2^19 bound the temporaries (135 MB).
static int32_t double_with_basepoint_affine(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* scalars64, u64* out, u8* inf,
                                            size_t n) {
    constexpr size_t SUB = (size_t)1 << 19;
    const size_t cap = n < SUB ? (n + 255) / 256 * 256 : SUB;
    if (int32_t rc = ensure_slot(ctx, ctx->pair_tmp, cap * 258, ctx->stream)) return rc;
    u8* base = static_cast<u8*>(ctx->pair_tmp.p);
    u8* a32 = base;
    u8* b32 = a32 + cap * 32;
    u64* ap = reinterpret_cast<u64*>(b32 + cap * 32);
    u64* bg = ap + cap * 12;
    u8* ap_inf = reinterpret_cast<u8*>(bg + cap * 12);
    u8* bg_inf = ap_inf + cap;
    const size_t wpb = ManyScratch::words(1, 2);
    const int grid_many = lane_grid(ctx, n < SUB ? n : SUB, CH_MANY_WAVES);
    if (int32_t rc = ensure_scratch(ctx, (size_t)grid_many * wpb)) return rc;
    const size_t in_w = form_limbs(in_form);
    for (size_t off = 0; off < n; off += SUB) {
        const size_t cnt = n - off < SUB ? n - off : SUB;
        const u64* pts_o = pts + off * in_w;
        const u8* sc_o = scalars64 + off * 64;
        hipLaunchKernelGGL(k_split_scalar_pairs, dim3(ew_grid(ctx, cnt * 8)), dim3(BLOCK), 0, ctx->stream,
                           reinterpret_cast<const u64*>(sc_o), reinterpret_cast<u64*>(a32), reinterpret_cast<u64*>(b32), cnt);
        if (int32_t rc = multiply_impl<true>(ctx, pts_o, in_form, a32, ap, ap_inf, cnt, CHEETAH_AFFINE)) return rc;
        if (int32_t rc = fixed_base_impl<false>(ctx, TAB(ctx, basepoint), TAB(ctx, wide), b32, bg, bg_inf, cnt, CHEETAH_AFFINE)) return rc;
        hipLaunchKernelGGL(k_affine_sum, dim3(ew_grid(ctx, cnt)), dim3(BLOCK), 0, ctx->stream, ap, ap_inf, bg, bg_inf, out + off * 12,
                           inf + off, cnt);
#define FIXUP(I) \
    hipLaunchKernelGGL((k_multiply_many_vartime<MJac, I, 0, true>), dim3(grid_many), dim3(BLOCK), 0, ctx->stream, ctx->tables, pts_o, sc_o, \
                       out + off * 12, inf + off, cnt, 1, 1, ctx->scratch, wpb)
        if (in_form == CHEETAH_AFFINE) FIXUP(0);
        else if (in_form == CHEETAH_PROJECTIVE) FIXUP(1);
        else FIXUP(2);
#undef FIXUP
    }
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

template <bool VARTIME>
static int32_t multiply_many_impl(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* scalars32, int m, int with_basepoint,
                                  u64* out, u8* inf, size_t n_groups, int out_form) {
    SET_DEVICE(ctx);
    if (out_form == CHEETAH_AFFINE && !inf) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    if (!aligned8(pts) || !aligned8(scalars32) || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
    if constexpr (VARTIME)
        if (with_basepoint && m == 1 && out_form == CHEETAH_AFFINE && !ctx->force_reference_schedule)
            return double_with_basepoint_affine(ctx, pts, in_form, scalars32, out, inf, n_groups);
    const int ns = m + (with_basepoint ? 1 : 0);
    int grid = lane_grid(ctx, n_groups, CH_MANY_WAVES);
    const size_t wpb = ManyScratch::words(m, ns);
    if (int32_t rc = ensure_scratch(ctx, (size_t)grid * wpb)) return rc;
    const size_t in_w = form_limbs(in_form), out_w = form_limbs(out_form);
    size_t rounds = (size_t)ctx->launch_rounds / (size_t)m;
    if (rounds == 0) rounds = 1;
    const size_t per_launch = (size_t)grid * BLOCK * rounds;
    u8* flags = nullptr;
    if (VARTIME && out_form == CHEETAH_AFFINE && !with_basepoint && !ctx->force_reference_schedule) {
        if (int32_t rc = ensure_slot(ctx, ctx->redo_flags, per_launch / 64 + 1, ctx->stream)) return rc;
        flags = static_cast<u8*>(ctx->redo_flags.p);
    }
    for (size_t off = 0; off < n_groups; off += per_launch) {
        const size_t cnt = n_groups - off < per_launch ? n_groups - off : per_launch;
        const u64* pts_o = pts + off * m * in_w;
        const u8* sc_o = scalars32 + off * ns * 32;
        u64* out_o = out + off * out_w;
        u8* inf_o = inf ? inf + off : nullptr;
#define LAUNCH(A, I, O)                                                                                                           \
    do {                                                                                                                          \
        if constexpr (VARTIME && O == 0) {                                                                                        \
            if (!with_basepoint && !ctx->force_reference_schedule) { /* affine results: indexed fixed-window ladder + fix-up */    \
                hipLaunchKernelGGL((k_multiply_many_ct<MJac, I, 0, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->tables,   \
                                   pts_o, sc_o, out_o, inf_o, cnt, m, ctx->scratch, wpb, flags);                                  \
                hipLaunchKernelGGL((k_multiply_many_vartime<MJac, I, 0, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream,           \
                                   ctx->tables, pts_o, sc_o, out_o, inf_o, cnt, m, 0, ctx->scratch, wpb, flags);                  \
                break;                                                                                                            \
            }                                                                                                                     \
        }                                                                                                                         \
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

template <bool VARTIME>
static int32_t multiply_many_op(cheetah_ctx* ctx, bool host, const u64* pts, int in_form, const u8* scalars32, int m, int with_basepoint,
                                u64* out, u8* inf, size_t n_groups, int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    if (m < 1 || m > CHEETAH_MAX_MANY) return fail(ctx, CHEETAH_ERR_INVALID, "m must be in [1, CHEETAH_MAX_MANY]");
    const size_t ns = (size_t)m + (with_basepoint ? 1 : 0);
    Op op;
    op.n = n_groups;
    op.chunk = (((size_t)1 << 18) / (size_t)m + 255) / 256 * 256;
    op.io = {in_(pts, (size_t)m * form_limbs(in_form) * 8), in_(scalars32, ns * 32), out_(out, form_limbs(out_form) * 8),
             out_(inf, 1, out_form != CHEETAH_AFFINE)};
    op.launch = [in_form, out_form, m, with_basepoint](cheetah_ctx* c, void* const* d, size_t cnt) {
        return multiply_many_impl<VARTIME>(c, at<u64>(d, 0), in_form, at<u8>(d, 1), m, with_basepoint, at<u64>(d, 2), at<u8>(d, 3), cnt,
                                           out_form);
    };
    return run_op(ctx, host, op);
}

extern "C" int32_t cheetah_multiply_many_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32, int m,
                                                   uint64_t* out, uint8_t* inf, size_t n_groups, int out_form) {
    return multiply_many_op<false>(ctx, false, pts, in_form, scalars32, m, 0, out, inf, n_groups, out_form);
}
extern "C" int32_t cheetah_multiply_many_vartime_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form,
                                                           const uint8_t* scalars32, int m, uint64_t* out, uint8_t* inf,
                                                           size_t n_groups, int out_form) {
    return multiply_many_op<true>(ctx, false, pts, in_form, scalars32, m, 0, out, inf, n_groups, out_form);
}
extern "C" int32_t cheetah_multiply_double_with_basepoint_vartime_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form,
                                                                            const uint8_t* scalars64, uint64_t* out, uint8_t* inf,
                                                                            size_t n, int out_form) {
    return multiply_many_op<true>(ctx, false, pts, in_form, scalars64, 1, 1, out, inf, n, out_form);
}
extern "C" int32_t cheetah_multiply_many_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32, int m,
                                               uint64_t* out, uint8_t* inf, size_t n_groups, int out_form) {
    return multiply_many_op<false>(ctx, true, pts, in_form, scalars32, m, 0, out, inf, n_groups, out_form);
}
extern "C" int32_t cheetah_multiply_many_vartime_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, const uint8_t* scalars32,
                                                       int m, uint64_t* out, uint8_t* inf, size_t n_groups, int out_form) {
    return multiply_many_op<true>(ctx, true, pts, in_form, scalars32, m, 0, out, inf, n_groups, out_form);
}
extern "C" int32_t cheetah_multiply_double_with_basepoint_vartime_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form,
                                                                        const uint8_t* scalars64, uint64_t* out, uint8_t* inf,
                                                                        size_t n, int out_form) {
    return multiply_many_op<true>(ctx, true, pts, in_form, scalars64, 1, 1, out, inf, n, out_form);
}

// One fixed scalar for the whole batch through the vartime ladder (every lane has the
// same NAF, so the wave never diverges).  out_form 3 = identity flag only; 4 = flags[i] &= is_identity.
static int32_t fixed_scalar_impl(cheetah_ctx* ctx, const u64* pts, int in_form, const u8* dev_scalar, u64* out, u8* inf, size_t n,
                                 int out_form) {
    SET_DEVICE(ctx);
    if (!aligned8(pts) || (out && !aligned8(out))) return fail(ctx, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
    int grid = lane_grid(ctx, n, CH_VT_WAVES);
    if (int32_t rc = ensure_scratch(ctx, (size_t)grid * LaneScratch::WORDS_PER_BLOCK)) return rc;
    const size_t in_w = form_limbs(in_form), out_w = out_form == CHEETAH_AFFINE ? 12 : 18;
    const size_t per_launch = (size_t)grid * BLOCK * ctx->launch_rounds;
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

static int32_t is_torsion_free_op(cheetah_ctx* ctx, bool host, const u64* pts, int in_form, u8* flags, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, in_form, CHEETAH_AFFINE)) return rc;
    Op op;
    op.n = n;
    op.chunk = (size_t)1 << 18;
    op.io = {in_(pts, form_limbs(in_form) * 8), out_(flags, 1)};
    op.launch = [in_form](cheetah_ctx* c, void* const* d, size_t cnt) {
        return fixed_scalar_impl(c, at<u64>(d, 0), in_form, c->fixed_scalars, nullptr, at<u8>(d, 1), cnt, 3);
    };
    return run_op(ctx, host, op);
}
static int32_t clear_cofactor_op(cheetah_ctx* ctx, bool host, const u64* pts, int in_form, u64* out, u8* inf, size_t n, int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, in_form, out_form)) return rc;
    Op op;
    op.n = n;
    op.chunk = (size_t)1 << 18;
    op.io = {in_(pts, form_limbs(in_form) * 8), out_(out, form_limbs(out_form) * 8), out_(inf, 1, out_form != CHEETAH_AFFINE)};
    op.launch = [in_form, out_form](cheetah_ctx* c, void* const* d, size_t cnt) {
        return fixed_scalar_impl(c, at<u64>(d, 0), in_form, c->fixed_scalars + 32, at<u64>(d, 1), at<u8>(d, 2), cnt, out_form);
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_is_torsion_free_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint8_t* flags, size_t n) {
    return is_torsion_free_op(ctx, false, pts, in_form, flags, n);
}
extern "C" int32_t cheetah_is_torsion_free_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint8_t* flags, size_t n) {
    return is_torsion_free_op(ctx, true, pts, in_form, flags, n);
}
extern "C" int32_t cheetah_clear_cofactor_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint64_t* out, uint8_t* inf,
                                                    size_t n, int out_form) {
    return clear_cofactor_op(ctx, false, pts, in_form, out, inf, n, out_form);
}
extern "C" int32_t cheetah_clear_cofactor_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint64_t* out, uint8_t* inf,
                                                size_t n, int out_form) {
    return clear_cofactor_op(ctx, true, pts, in_form, out, inf, n, out_form);
}

static int32_t is_on_curve_op(cheetah_ctx* ctx, bool host, const u64* pts, int in_form, u8* flags, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, in_form, CHEETAH_AFFINE)) return rc;
    Op op;
    op.n = n;
    op.io = {in_(pts, form_limbs(in_form) * 8), out_(flags, 1)};
    op.launch = [in_form](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned8(d[0])) return fail(c, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
        const int grid = ew_grid(c, cnt);
        if (in_form == CHEETAH_AFFINE)
            hipLaunchKernelGGL((k_is_on_curve<0>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u8>(d, 1), cnt);
        else if (in_form == CHEETAH_PROJECTIVE)
            hipLaunchKernelGGL((k_is_on_curve<1>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u8>(d, 1), cnt);
        else
            hipLaunchKernelGGL((k_is_on_curve<2>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u8>(d, 1), cnt);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_is_on_curve_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint8_t* flags, size_t n) {
    return is_on_curve_op(ctx, false, pts, in_form, flags, n);
}
extern "C" int32_t cheetah_is_on_curve_batch(cheetah_ctx* ctx, const uint64_t* pts, int in_form, uint8_t* flags, size_t n) {
    return is_on_curve_op(ctx, true, pts, in_form, flags, n);
}

// Complete additions and doublings (elementwise).
static int32_t add_op(cheetah_ctx* ctx, bool host, const u64* lhs, int lhs_form, const u64* rhs, int rhs_form, u64* out, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (lhs_form != CHEETAH_PROJECTIVE && lhs_form != CHEETAH_JACOBIAN)
        return fail(ctx, CHEETAH_ERR_INVALID, "lhs_form must be CHEETAH_PROJECTIVE or CHEETAH_JACOBIAN");
    if (rhs_form != CHEETAH_AFFINE && rhs_form != lhs_form)
        return fail(ctx, CHEETAH_ERR_INVALID, "rhs_form must be CHEETAH_AFFINE (mixed addition) or equal to lhs_form");
    Op op;
    op.n = n;
    op.io = {in_(lhs, 18 * 8), in_(rhs, form_limbs(rhs_form) * 8), out_(out, 18 * 8)};
    op.launch = [lhs_form, rhs_form](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned8(d[0]) || !aligned8(d[1]) || !aligned8(d[2]))
            return fail(c, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
        const int grid = ew_grid(c, cnt);
#define LAUNCH(L, R) hipLaunchKernelGGL((k_add<L, R>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), at<u64>(d, 2), cnt)
        if (lhs_form == CHEETAH_PROJECTIVE) {
            if (rhs_form == CHEETAH_AFFINE) LAUNCH(1, 0);
            else LAUNCH(1, 1);
        } else {
            if (rhs_form == CHEETAH_AFFINE) LAUNCH(2, 0);
            else LAUNCH(2, 2);
        }
#undef LAUNCH
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
static int32_t double_op(cheetah_ctx* ctx, bool host, const u64* pts, int form, u32 k, u64* out, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (form != CHEETAH_PROJECTIVE && form != CHEETAH_JACOBIAN)
        return fail(ctx, CHEETAH_ERR_INVALID, "form must be CHEETAH_PROJECTIVE or CHEETAH_JACOBIAN");
    if (k < 1) return fail(ctx, CHEETAH_ERR_INVALID, "double_multi needs at least one doubling (the reference asserts n >= 1)");
    Op op;
    op.n = n;
    op.io = {in_(pts, 18 * 8), out_(out, 18 * 8)};
    op.launch = [form, k](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned8(d[0]) || !aligned8(d[1])) return fail(c, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
        const int grid = ew_grid(c, cnt);
        if (form == CHEETAH_PROJECTIVE)
            hipLaunchKernelGGL((k_double<1>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), cnt, k);
        else
            hipLaunchKernelGGL((k_double<2>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), cnt, k);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
static int32_t unchecked_op(cheetah_ctx* ctx, bool host, const u64* lhs, int form, const u64* rhs_xy, u32 k, u64* out, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (form != CHEETAH_PROJECTIVE && form != CHEETAH_JACOBIAN)
        return fail(ctx, CHEETAH_ERR_INVALID, "form must be CHEETAH_PROJECTIVE or CHEETAH_JACOBIAN");
    const bool add = rhs_xy != nullptr;
    if (!add && k < 1) return fail(ctx, CHEETAH_ERR_INVALID, "double_multi_unchecked needs at least one doubling (the reference asserts n >= 1)");
    Op op;
    op.n = n;
    if (add) op.io = {in_(lhs, 18 * 8), in_(rhs_xy, 12 * 8), out_(out, 18 * 8)};
    else op.io = {in_(lhs, 18 * 8), out_(out, 18 * 8)};
    op.launch = [form, add, k](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned8(d[0]) || !aligned8(d[1]) || (add && !aligned8(d[2]))) return fail(c, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
        const int grid = ew_grid(c, cnt);
        if (add && form == CHEETAH_PROJECTIVE)
            hipLaunchKernelGGL((k_add_mixed_unchecked<1>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), at<u64>(d, 2), cnt);
        else if (add)
            hipLaunchKernelGGL((k_add_mixed_unchecked<2>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), at<u64>(d, 2), cnt);
        else if (form == CHEETAH_PROJECTIVE)
            hipLaunchKernelGGL((k_double_unchecked<1>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), cnt, k);
        else
            hipLaunchKernelGGL((k_double_unchecked<2>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), cnt, k);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
static int32_t neg_op(cheetah_ctx* ctx, bool host, const u64* pts, int form, u64* out, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, form, form)) return rc;
    Op op;
    op.n = n;
    op.io = {in_(pts, form_limbs(form) * 8), out_(out, form_limbs(form) * 8)};
    op.launch = [form](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned8(d[0]) || !aligned8(d[1])) return fail(c, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
        const int grid = ew_grid(c, cnt * 3);
        if (form == CHEETAH_AFFINE)
            hipLaunchKernelGGL((k_neg<0>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), cnt);
        else  // ProjectivePoint::neg and JacobianPoint::neg are the same map (x, -y, z)
            hipLaunchKernelGGL((k_neg<1>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), cnt);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
static int32_t eq_op(cheetah_ctx* ctx, bool host, const u64* a, const u64* b, int form, u8* flags, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, form, form)) return rc;
    Op op;
    op.n = n;
    op.io = {in_(a, form_limbs(form) * 8), in_(b, form_limbs(form) * 8), out_(flags, 1)};
    op.launch = [form](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned8(d[0]) || !aligned8(d[1])) return fail(c, CHEETAH_ERR_INVALID, "device pointers must be 8-byte aligned");
        const int grid = ew_grid(c, cnt);
#define LAUNCH(F) hipLaunchKernelGGL((k_point_eq<F>), dim3(grid), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), at<u8>(d, 2), cnt)
        if (form == CHEETAH_AFFINE) LAUNCH(0);
        else if (form == CHEETAH_PROJECTIVE) LAUNCH(1);
        else LAUNCH(2);
#undef LAUNCH
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_add_batch_dev(cheetah_ctx* ctx, const uint64_t* lhs, int lhs_form, const uint64_t* rhs, int rhs_form,
                                         uint64_t* out, size_t n) {
    return add_op(ctx, false, lhs, lhs_form, rhs, rhs_form, out, n);
}
extern "C" int32_t cheetah_add_batch(cheetah_ctx* ctx, const uint64_t* lhs, int lhs_form, const uint64_t* rhs, int rhs_form,
                                     uint64_t* out, size_t n) {
    return add_op(ctx, true, lhs, lhs_form, rhs, rhs_form, out, n);
}
extern "C" int32_t cheetah_double_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int form, uint64_t* out, size_t n) {
    return double_op(ctx, false, pts, form, 1, out, n);
}
extern "C" int32_t cheetah_double_batch(cheetah_ctx* ctx, const uint64_t* pts, int form, uint64_t* out, size_t n) {
    return double_op(ctx, true, pts, form, 1, out, n);
}
extern "C" int32_t cheetah_double_multi_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int form, uint32_t k, uint64_t* out, size_t n) {
    return double_op(ctx, false, pts, form, k, out, n);
}
extern "C" int32_t cheetah_double_multi_batch(cheetah_ctx* ctx, const uint64_t* pts, int form, uint32_t k, uint64_t* out, size_t n) {
    return double_op(ctx, true, pts, form, k, out, n);
}
extern "C" int32_t cheetah_double_multi_unchecked_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int form, uint32_t k, uint64_t* out,
                                                           size_t n) {
    return unchecked_op(ctx, false, pts, form, nullptr, k, out, n);
}
extern "C" int32_t cheetah_double_multi_unchecked_batch(cheetah_ctx* ctx, const uint64_t* pts, int form, uint32_t k, uint64_t* out, size_t n) {
    return unchecked_op(ctx, true, pts, form, nullptr, k, out, n);
}
extern "C" int32_t cheetah_add_mixed_unchecked_batch_dev(cheetah_ctx* ctx, const uint64_t* lhs, int form, const uint64_t* rhs_xy,
                                                         uint64_t* out, size_t n) {
    if (ctx && !rhs_xy && n) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    return unchecked_op(ctx, false, lhs, form, rhs_xy, 0, out, n);
}
extern "C" int32_t cheetah_add_mixed_unchecked_batch(cheetah_ctx* ctx, const uint64_t* lhs, int form, const uint64_t* rhs_xy, uint64_t* out,
                                                     size_t n) {
    if (ctx && !rhs_xy && n) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    return unchecked_op(ctx, true, lhs, form, rhs_xy, 0, out, n);
}
extern "C" int32_t cheetah_neg_batch_dev(cheetah_ctx* ctx, const uint64_t* pts, int form, uint64_t* out, size_t n) {
    return neg_op(ctx, false, pts, form, out, n);
}
extern "C" int32_t cheetah_neg_batch(cheetah_ctx* ctx, const uint64_t* pts, int form, uint64_t* out, size_t n) {
    return neg_op(ctx, true, pts, form, out, n);
}
extern "C" int32_t cheetah_eq_batch_dev(cheetah_ctx* ctx, const uint64_t* a, const uint64_t* b, int form, uint8_t* flags, size_t n) {
    return eq_op(ctx, false, a, b, form, flags, n);
}
extern "C" int32_t cheetah_eq_batch(cheetah_ctx* ctx, const uint64_t* a, const uint64_t* b, int form, uint8_t* flags, size_t n) {
    return eq_op(ctx, true, a, b, form, flags, n);
}

// -------------------------------------------------------------- encodings -----

template <int REC>
static int32_t encode_op(cheetah_ctx* ctx, bool host, const u64* xy, const u8* inf, u8* out, size_t n) {
    Op op;
    op.n = n;
    op.io = {in_(xy, 96), in_(inf, 1), out_(out, REC)};
    op.launch = [](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned16(d[0]) || !aligned4(d[2])) return fail(c, CHEETAH_ERR_INVALID, "xy must be 16-byte aligned, the byte array 4-byte aligned");
        hipLaunchKernelGGL((k_encode<REC>), dim3(ew_grid(c, cnt)), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u8>(d, 1), at<u8>(d, 2), cnt);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_encode_uncompressed_batch_dev(cheetah_ctx* ctx, const uint64_t* xy, const uint8_t* inf, uint8_t* out97, size_t n) {
    return encode_op<97>(ctx, false, xy, inf, out97, n);
}
extern "C" int32_t cheetah_encode_compressed_batch_dev(cheetah_ctx* ctx, const uint64_t* xy, const uint8_t* inf, uint8_t* out49, size_t n) {
    return encode_op<49>(ctx, false, xy, inf, out49, n);
}
extern "C" int32_t cheetah_encode_uncompressed_batch(cheetah_ctx* ctx, const uint64_t* xy, const uint8_t* inf, uint8_t* out97, size_t n) {
    return encode_op<97>(ctx, true, xy, inf, out97, n);
}
extern "C" int32_t cheetah_encode_compressed_batch(cheetah_ctx* ctx, const uint64_t* xy, const uint8_t* inf, uint8_t* out49, size_t n) {
    return encode_op<49>(ctx, true, xy, inf, out49, n);
}

template <int REC>
static int32_t decode_op(cheetah_ctx* ctx, bool host, const u8* in, u64* xy, u8* inf, u8* is_some, size_t n, int checked) {
    Op op;
    op.n = n;
    op.chunk = checked ? (size_t)1 << 18 : (size_t)1 << 20;
    op.io = {in_(in, REC), out_(xy, 96), out_(inf, 1), out_(is_some, 1)};
    op.launch = [checked](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        const u8* src = at<u8>(d, 0);
        u64* pxy = at<u64>(d, 1);
        u8 *pinf = at<u8>(d, 2), *psome = at<u8>(d, 3);
        if (!aligned16(pxy) || !aligned4(src)) return fail(c, CHEETAH_ERR_INVALID, "xy must be 16-byte aligned, the byte array 4-byte aligned");
        const int grid = ew_grid(c, cnt);
        if (REC == 97) {
            if (checked)
                hipLaunchKernelGGL((k_decode_uncompressed<1>), dim3(grid), dim3(BLOCK), 0, c->stream, src, pxy, pinf, psome, cnt);
            else
                hipLaunchKernelGGL((k_decode_uncompressed<0>), dim3(grid), dim3(BLOCK), 0, c->stream, src, pxy, pinf, psome, cnt);
        } else {
            hipLaunchKernelGGL(k_decode_compressed, dim3(grid), dim3(BLOCK), 0, c->stream, src, pxy, pinf, psome, cnt);
        }
        HIP_TRY(c, hipGetLastError());
        if (checked) {  // is_some &= is_torsion_free, then rejected records leave as (0, 0, inf = 0)
            if (int32_t rc = fixed_scalar_impl(c, pxy, CHEETAH_AFFINE, c->fixed_scalars, nullptr, psome, cnt, 4)) return rc;
            hipLaunchKernelGGL(k_zero_rejected, dim3(grid), dim3(BLOCK), 0, c->stream, pxy, pinf, psome, cnt);
            HIP_TRY(c, hipGetLastError());
        }
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_decode_uncompressed_batch_dev(cheetah_ctx* ctx, const uint8_t* in97, uint64_t* xy, uint8_t* inf,
                                                         uint8_t* is_some, size_t n, int checked) {
    return decode_op<97>(ctx, false, in97, xy, inf, is_some, n, checked);
}
extern "C" int32_t cheetah_decode_compressed_batch_dev(cheetah_ctx* ctx, const uint8_t* in49, uint64_t* xy, uint8_t* inf,
                                                       uint8_t* is_some, size_t n, int checked) {
    return decode_op<49>(ctx, false, in49, xy, inf, is_some, n, checked);
}
extern "C" int32_t cheetah_decode_uncompressed_batch(cheetah_ctx* ctx, const uint8_t* in97, uint64_t* xy, uint8_t* inf, uint8_t* is_some,
                                                     size_t n, int checked) {
    return decode_op<97>(ctx, true, in97, xy, inf, is_some, n, checked);
}
extern "C" int32_t cheetah_decode_compressed_batch(cheetah_ctx* ctx, const uint8_t* in49, uint64_t* xy, uint8_t* inf, uint8_t* is_some,
                                                   size_t n, int checked) {
    return decode_op<49>(ctx, true, in49, xy, inf, is_some, n, checked);
}

static int32_t affine_from_x_op(cheetah_ctx* ctx, bool host, const u64* x, u64* xy, u8* inf, u8* is_some, size_t n) {
    Op op;
    op.n = n;
    op.io = {in_(x, 48), out_(xy, 96), out_(inf, 1), out_(is_some, 1)};
    op.launch = [](cheetah_ctx* c, void* const* d, size_t cnt) -> int32_t {
        SET_DEVICE(c);
        if (!aligned16(d[0]) || !aligned16(d[1])) return fail(c, CHEETAH_ERR_INVALID, "Fp6 device arrays must be 16-byte aligned");
        hipLaunchKernelGGL(k_affine_from_x, dim3(ew_grid(c, cnt)), dim3(BLOCK), 0, c->stream, at<u64>(d, 0), at<u64>(d, 1), at<u8>(d, 2),
                           at<u8>(d, 3), cnt);
        HIP_TRY(c, hipGetLastError());
        return CHEETAH_OK;
    };
    return run_op(ctx, host, op);
}
extern "C" int32_t cheetah_affine_from_x_batch_dev(cheetah_ctx* ctx, const uint64_t* x, uint64_t* xy, uint8_t* inf, uint8_t* is_some,
                                                   size_t n) {
    return affine_from_x_op(ctx, false, x, xy, inf, is_some, n);
}
extern "C" int32_t cheetah_affine_from_x_batch(cheetah_ctx* ctx, const uint64_t* x, uint64_t* xy, uint8_t* inf, uint8_t* is_some, size_t n) {
    return affine_from_x_op(ctx, true, x, xy, inf, is_some, n);
}

// ------------------------------------------------------- tables, synth -----

extern "C" int32_t cheetah_read_table(cheetah_ctx* ctx, int which, uint64_t* out_xy, size_t max_points) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    SET_DEVICE(ctx);
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
    if (!ctx) return CHEETAH_ERR_INVALID;
    SET_DEVICE(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!out || !aligned8(out)) return fail(ctx, CHEETAH_ERR_INVALID, "bad pointer");
    hipLaunchKernelGGL(k_synth_fp6, dim3(ctx->num_cus * 8), dim3(BLOCK), 0, ctx->stream, seed, start, out, n * 6);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_synth_scalars_dev(cheetah_ctx* ctx, uint64_t seed, uint64_t start, uint8_t* out32, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    SET_DEVICE(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!out32 || !aligned8(out32)) return fail(ctx, CHEETAH_ERR_INVALID, "bad pointer");
    hipLaunchKernelGGL(k_synth_scalars, dim3(ctx->num_cus * 8), dim3(BLOCK), 0, ctx->stream, seed, start,
                       reinterpret_cast<u64*>(out32), n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

// {Affine,Projective,Jacobian}Point::random with the seeded generator in the rng's place: candidates by
// k_random_candidates (x drawn, y = +-sqrt(x^3 + x + B)), then clear_cofactor through the fixed-scalar ladder, whose
// representative is the one the reference's clear_cofactor returns.  Sub-batches of 2^19 bound the temporaries (48 MB).
// Runs on the context's first device (like the other synthetic-input generators).
extern "C" int32_t cheetah_random_points_dev(cheetah_ctx* ctx, uint64_t seed, uint64_t start, uint64_t* out, uint8_t* inf, size_t n,
                                             int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, CHEETAH_AFFINE, out_form)) return rc;
    SET_DEVICE(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!out || !aligned16(out) || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "bad pointer");
    constexpr size_t SUB = (size_t)1 << 19;
    const size_t cap = n < SUB ? n : SUB;
    if (int32_t rc = ensure_slot(ctx, ctx->pair_tmp, cap * 258, ctx->stream)) return rc;
    u64* cand = static_cast<u64*>(ctx->pair_tmp.p);
    const size_t out_w = form_limbs(out_form);
    for (size_t off = 0; off < n; off += SUB) {
        const size_t cnt = n - off < SUB ? n - off : SUB;
        hipLaunchKernelGGL(k_random_candidates, dim3(ew_grid(ctx, cnt)), dim3(BLOCK), 0, ctx->stream, seed, start + off, cand, cnt);
        if (int32_t rc = fixed_scalar_impl(ctx, cand, CHEETAH_AFFINE, ctx->fixed_scalars + 32, out + off * out_w, inf ? inf + off : nullptr, cnt,
                                           out_form))
            return rc;
    }
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

// The same stream into HOST arrays (what a host-side `random()` replacement calls): generated on the first device in the
// same sub-batches and copied out; element i of a stream does not depend on how the stream is cut.
extern "C" int32_t cheetah_random_points(cheetah_ctx* ctx, uint64_t seed, uint64_t start, uint64_t* out, uint8_t* inf, size_t n,
                                         int out_form) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (int32_t rc = check_forms(ctx, CHEETAH_AFFINE, out_form)) return rc;
    SET_DEVICE(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!out || (out_form == CHEETAH_AFFINE && !inf)) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    constexpr size_t SUB = (size_t)1 << 19;
    const size_t cap = n < SUB ? n : SUB, out_w = form_limbs(out_form);
    u64* d_out = nullptr;
    u8* d_inf = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), cap * out_w * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_inf), cap);
    int32_t rc = CHEETAH_OK;
    for (size_t off = 0; off < n && e == hipSuccess && rc == CHEETAH_OK; off += SUB) {
        const size_t cnt = n - off < SUB ? n - off : SUB;
        rc = cheetah_random_points_dev(ctx, seed, start + off, d_out, d_inf, cnt, out_form);
        if (rc != CHEETAH_OK) break;
        e = hipMemcpyAsync(out + off * out_w, d_out, cnt * out_w * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && inf) e = hipMemcpyAsync(inf + off, d_inf, cnt, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (d_out) (void)hipFree(d_out);
    if (d_inf) (void)hipFree(d_inf);
    if (rc != CHEETAH_OK) return rc;
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? CHEETAH_ERR_NOMEM : CHEETAH_ERR_HIP, hipGetErrorString(e));
    return CHEETAH_OK;
}

// ---------------------------------------------------------- diagnostics -----

extern "C" int32_t cheetah_debug_op_dev(cheetah_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    SET_DEVICE(ctx);
    if (n == 0) return CHEETAH_OK;
    if (!a || !out) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    hipLaunchKernelGGL(k_debug_op, dim3(lane_grid(ctx, n, 4)), dim3(BLOCK), 0, ctx->stream, op, a, b, out, n);
    HIP_TRY(ctx, hipGetLastError());
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_multiply_units_per_launch(cheetah_ctx* ctx, size_t n, int vartime, int out_form, size_t* units) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    if (!units) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    const bool pooled = vartime && out_form != CHEETAH_AFFINE;
    const size_t slots = pooled ? VtPool::SLOTS : 1;
    const size_t per_launch = (size_t)lane_grid(ctx, ((n ? n : 1) + slots - 1) / slots, pooled ? CH_VT_POOL_WAVES : CH_CT_WAVES) * BLOCK * slots * ctx->launch_rounds * (pooled ? POOL_ROUNDS_FACTOR : 1);
    *units = n < per_launch ? n : per_launch;
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_calibrate_mad_peak(cheetah_ctx* ctx, double* tera_mads_per_s) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    SET_DEVICE(ctx);
    if (!tera_mads_per_s) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    u64* d = ctx->scratch;  // the kernel stores at most one word, and only for a value that does not occur
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
    *tera_mads_per_s = (double)blocks * BLOCK * 64.0 * iters / (best * 1e-3) / 1e12;
    return CHEETAH_OK;
}

extern "C" int32_t cheetah_timer_start(cheetah_ctx* ctx) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    SET_DEVICE(ctx);
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return CHEETAH_OK;
}
extern "C" int32_t cheetah_timer_stop_ms(cheetah_ctx* ctx, float* ms) {
    if (!ctx) return CHEETAH_ERR_INVALID;
    SET_DEVICE(ctx);
    if (!ms) return fail(ctx, CHEETAH_ERR_INVALID, "null pointer");
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return CHEETAH_OK;
}
