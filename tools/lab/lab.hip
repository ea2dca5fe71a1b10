// Kernel lab: ONE kernel family of csrc/kernels.hip.h per build, behind a tiny C interface, so that a variant compiles in
// seconds instead of the minutes the whole library takes.  Inputs and reference outputs come from the product library
// (Python: cheetah_amd.Engine); this file only launches the kernel under test on the caller's device pointers.
//   tools/lab/build.sh <name> -DLAB_BN [-D...]   ->  tools/lab/lib<name>.so
// Not part of the product: nothing in cheetah_amd/ loads it.
#include "../../cheetah_amd/csrc/kernels.hip.h"
using namespace cheetah;

extern "C" int lab_num_cus() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
    return p.multiProcessorCount;
}

#ifdef LAB_BN
extern "C" int lab_batch_normalize_jac(const u64* xyz, u64* xy, u8* inf, size_t n, void* stream) {
    const size_t nchunks = (n + CH_BN_CHUNK - 1) / CH_BN_CHUNK;
    size_t want = (nchunks + BLOCK - 1) / BLOCK, cap = (size_t)lab_num_cus() * CH_BN_WAVES;
    const int grid = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL((k_batch_normalize<1>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, xyz, xy, inf, n);
    return (int)hipGetLastError();
}
#endif

#ifdef LAB_VT
// ProjectivePoint / JacobianPoint::multiply_vartime with the reference's own representative (L2)
extern "C" int lab_multiply_vartime(const DeviceTables* tab, const u64* pts, const u8* scalars, u64* out, u8* inf, size_t n, u64* scratch,
                                    int blocks_per_cu, int proj, void* stream) {
    size_t want = (n + BLOCK - 1) / BLOCK, cap = (size_t)lab_num_cus() * blocks_per_cu;
    const int grid = (int)(want < cap ? want : cap);
    if (proj)
        hipLaunchKernelGGL((k_multiply_vartime<Proj, 0, 1, false>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, tab, pts, scalars, out, inf, n, scratch, 0);
    else
        hipLaunchKernelGGL((k_multiply_vartime<MJac, 0, 2, false>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, tab, pts, scalars, out, inf, n, scratch, 0);
    return (int)hipGetLastError();
}
extern "C" size_t lab_scratch_words_per_block() { return LaneScratch::WORDS_PER_BLOCK; }
extern "C" int lab_multiply_vartime_pool(const DeviceTables* tab, const u64* pts, const u8* scalars, u64* out, u8* inf, size_t n, u64* scratch,
                                         int blocks_per_cu, int proj, void* stream) {
    const size_t per_round = (size_t)VtPool::SLOTS * BLOCK;
    size_t want = (n + per_round - 1) / per_round, cap = (size_t)lab_num_cus() * blocks_per_cu;
    const int grid = (int)(want < cap ? want : cap);
    if (proj)
        hipLaunchKernelGGL((k_multiply_vartime_pool<Proj, 0, 1>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, tab, pts, scalars, out, inf, n, scratch, VtPool::words(3));
    else
        hipLaunchKernelGGL((k_multiply_vartime_pool<MJac, 0, 2>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, tab, pts, scalars, out, inf, n, scratch, VtPool::words(4));
    return (int)hipGetLastError();
}
extern "C" size_t lab_pool_words_per_block(int proj) { return VtPool::words(proj ? 3 : 4); }
#endif

#ifdef LAB_MANY
// multiply_many (constant time, vartime = 0) / its indexed-lookup form (vartime = 1), affine in, affine out
extern "C" int lab_multiply_many(const DeviceTables* tab, const u64* pts, const u8* scalars, u64* out, u8* inf, size_t n_groups, int m,
                                 u64* scratch, int vartime, void* stream) {
    size_t want = (n_groups + BLOCK - 1) / BLOCK, cap = (size_t)lab_num_cus() * CH_MANY_WAVES;
    const int grid = (int)(want < cap ? want : cap);
    const size_t wpb = ManyScratch::words(m, m);
    if (vartime)
        hipLaunchKernelGGL((k_multiply_many_ct<MJac, 0, 0, true>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, tab, pts, scalars, out, inf, n_groups, m, scratch, wpb);
    else
        hipLaunchKernelGGL((k_multiply_many_ct<MJac, 0, 0, false>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, tab, pts, scalars, out, inf, n_groups, m, scratch, wpb);
    return (int)hipGetLastError();
}
extern "C" size_t lab_many_scratch_words(int m) { return (size_t)lab_num_cus() * CH_MANY_WAVES * ManyScratch::words(m, m); }
#endif

#ifdef LAB_W5
// VERDICT r02 item 1b, measured: AffinePoint::multiply with AFFINE results by signed 5-bit windows -- 52 digits in
// [-16, 16) (the top one un-recentred), a 16-entry table by build_table_affine, 52 additions and 260 doublings instead
// of 64 and 256; everything else (accumulator, formulas, scan, launch shape) is k_multiply_ct<MJac, 0, 0, DIRECT>.
// The shift constants 2^5 S and -2^260 S come from k_lab_w5_consts.
CH_DEV void recode_radix32(const u64 k[4], i8* digits /* [52][BLOCK] */, int t) {
    int carry = 0;
#pragma unroll
    for (int i = 0; i < 52; i++) {
        const int pos = 5 * i, idx = pos >> 6, bit = pos & 63;
        const u64 lo = k[idx], hi = idx < 3 ? k[idx + 1] : 0ull;
        const u32 w = (u32)(bit <= 59 ? (lo >> bit) : ((lo >> bit) | (hi << (64 - bit)))) & 31u;
        int d = (int)w + carry;
        if (i < 51) {
            carry = (d + 16) >> 5;
            d -= carry << 5;
        }
        digits[i * BLOCK + t] = (i8)d;
    }
}
__global__ void k_lab_w5_consts(const DeviceTables* tab, u64* consts /* 2^5 S affine [12], -2^260 S affine [12] */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Proj cur = proj_from_affine(load_affine(tab->shift));
    for (int i = 0; i < 260; i++) {
        if (i == 5) store_affine(consts, nullptr, 0, affine_from_proj(cur));
        cur = proj_double(cur);
    }
    Affine m = affine_from_proj(cur);
    m.y = fp6_neg(m.y);
    store_affine(consts + 12, nullptr, 0, m);
}
template <bool DIRECT>
__global__ void __launch_bounds__(BLOCK, CH_CT_WAVES) k_lab_multiply_w5(const DeviceTables* __restrict__ tab, const u64* __restrict__ consts,
                                                                       const u64* __restrict__ pts, const u8* __restrict__ scalars,
                                                                       u64* __restrict__ out, u8* __restrict__ inf, size_t n,
                                                                       u64* __restrict__ scratch, int n_entries /* 16, a run-time value as in the product */) {
    __shared__ i8 s_digits[52 * BLOCK];
    const int t = threadIdx.x;
    u64* s_tab = scratch + (size_t)blockIdx.x * LaneScratch::WORDS_PER_BLOCK;
    u64* s_z = s_tab + (size_t)16 * 12 * BLOCK;
    u64* s_prefix = s_z + (size_t)16 * 6 * BLOCK;
    const MJac shift32 = mjac_from_affine(load_affine(consts));
    const Affine mshift = load_affine(consts + 12);
    const size_t nround = (n + BLOCK - 1) / BLOCK;
    for (size_t round = blockIdx.x; round < nround; round += gridDim.x) {
        const size_t i = round * BLOCK + t;
        const bool active = i < n;
        const size_t ii = active ? i : n - 1;
        u64 k[4];
        load_scalar(scalars, ii, k);
        __syncthreads();
        recode_radix32(k, s_digits, t);
        __syncthreads();
        const u32 ident = build_lookup_table<0>(pts, ii, n_entries, s_tab, s_z, s_prefix, t);
        int d = s_digits[51 * BLOCK + t];
        MJac acc = Ladder<MJac>::add(shift32, DIRECT ? index_lookup(s_tab, ident, d, t) : scan_lookup(s_tab, n_entries, ident, d, t));
        for (int j = 50; j >= 0; j--) {
            acc = Ladder<MJac>::dbls_then_add(acc, 5);
            d = s_digits[j * BLOCK + t];
            acc = Ladder<MJac>::add_after_dbl(acc, DIRECT ? index_lookup(s_tab, ident, d, t) : scan_lookup(s_tab, n_entries, ident, d, t));
        }
        acc = Ladder<MJac>::add_last(acc, mshift);
        if (active) emit_point<0>(out, inf, i, acc);
    }
}
// Upper bound of what a co-Z step (the window's last doubling and its addition sharing z, 11 M + 7 S per
// window instead of 11 M + 9 S) could gain, WITHOUT building it: the product's radix-16 ladder with the two squarings that
// rebuild z^4 after every addition simply left out.  The results are wrong (z4 is stale), the instruction stream is the
// product's minus 2 S per window and nothing added for the odd-digit recoding, the table of odd multiples or the closing
// correction a real co-Z ladder needs; the kernel is constant time, so its duration does not depend on the values.
__global__ void __launch_bounds__(BLOCK, CH_CT_WAVES) k_lab_multiply_w4_minus2s(const DeviceTables* __restrict__ tab, const u64* __restrict__ pts,
                                                                               const u8* __restrict__ scalars, u64* __restrict__ out,
                                                                               u8* __restrict__ inf, size_t n, u64* __restrict__ scratch, int n_entries /* 8 */) {
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
        const u32 ident = build_lookup_table<0>(pts, ii, n_entries, s_tab, s_z, s_prefix, t);
        int d = s_digits[63 * BLOCK + t];
        MJac acc = Ladder<MJac>::add(Ladder<MJac>::shift16(tab), scan_lookup(s_tab, n_entries, ident, d, t));
        for (int j = 62; j >= 0; j--) {
            acc = Ladder<MJac>::dbls_then_add(acc, 4);
            d = s_digits[j * BLOCK + t];
            acc = Ladder<MJac>::add_last(acc, scan_lookup(s_tab, n_entries, ident, d, t));  // no z^4: 2 S fewer, wrong on purpose
        }
        acc = Ladder<MJac>::add_last(acc, mshift);
        if (active) emit_point<0>(out, inf, i, acc);
    }
}
extern "C" int lab_w5_consts(const DeviceTables* tab, u64* consts) {
    hipLaunchKernelGGL(k_lab_w5_consts, dim3(1), dim3(64), 0, 0, tab, consts);
    return (int)hipDeviceSynchronize();
}
// w = 5: this file's kernel; w = 4: the product's k_multiply_ct, same launch shape, for the comparison on one box
extern "C" int lab_multiply_w(int w, int direct, const DeviceTables* tab, const u64* consts, const u64* pts, const u8* scalars, u64* out, u8* inf,
                              size_t n, u64* scratch, void* stream) {
    size_t want = (n + BLOCK - 1) / BLOCK, cap = (size_t)lab_num_cus() * CH_CT_WAVES;
    const int grid = (int)(want < cap ? want : cap);
    hipStream_t st = (hipStream_t)stream;
    if (w == 40) hipLaunchKernelGGL(k_lab_multiply_w4_minus2s, dim3(grid), dim3(BLOCK), 0, st, tab, pts, scalars, out, inf, n, scratch, 8);
    else if (w == 5 && direct) hipLaunchKernelGGL((k_lab_multiply_w5<true>), dim3(grid), dim3(BLOCK), 0, st, tab, consts, pts, scalars, out, inf, n, scratch, 16);
    else if (w == 5) hipLaunchKernelGGL((k_lab_multiply_w5<false>), dim3(grid), dim3(BLOCK), 0, st, tab, consts, pts, scalars, out, inf, n, scratch, 16);
    else if (direct) hipLaunchKernelGGL((k_multiply_ct<MJac, 0, 0, true>), dim3(grid), dim3(BLOCK), 0, st, tab, pts, scalars, out, inf, n, scratch);
    else hipLaunchKernelGGL((k_multiply_ct<MJac, 0, 0, false>), dim3(grid), dim3(BLOCK), 0, st, tab, pts, scalars, out, inf, n, scratch);
    return (int)hipGetLastError();
}
extern "C" size_t lab_ct_scratch_words() { return (size_t)lab_num_cus() * CH_CT_WAVES * LaneScratch::WORDS_PER_BLOCK; }
#endif

#ifdef LAB_RAND
extern "C" int lab_random_candidates(u64 seed, u64 start, u64* xy, size_t n, void* stream) {
    size_t want = (n + BLOCK - 1) / BLOCK, cap = (size_t)lab_num_cus() * 8;
    hipLaunchKernelGGL(k_random_candidates, dim3((int)(want < cap ? want : cap)), dim3(BLOCK), 0, (hipStream_t)stream, seed, start, xy, n);
    return (int)hipGetLastError();
}
#endif

// The reference's lazy_static tables (shift points, BASEPOINT_LOOKUP, ODD_MULTIPLES) for the kernels under test: built by
// the product's own k_build_tables from the generator and the shift point (device pointers to 12 limbs each).
extern "C" void* lab_build_tables(const u64* gen_xy, const u64* shift_xy) {
    DeviceTables* tab = nullptr;
    if (hipMalloc(&tab, sizeof(DeviceTables)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(k_build_tables, dim3(1), dim3(BUILD_TABLES_THREADS), 0, 0, tab, gen_xy, shift_xy);
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    return tab;
}

#ifdef LAB_COUNT
extern "C" int lab_read_counters(unsigned long long* out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(lab_counters), 32) != hipSuccess) return 1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lab_counters), z, 32) != hipSuccess) return 1; }
    return 0;
}
#endif
