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
