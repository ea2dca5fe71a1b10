// Memory-access pattern probe for the AoS point records of batch_normalize (144-byte records in, 96 + 1 bytes out):
// how fast can a kernel with NO arithmetic move them with (a) one record per lane, 16-byte accesses at a 144-byte
// stride (what k_batch_normalize does) and (b) wave-contiguous 1 KiB accesses?  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int BLOCK = 256;

// (a) lane = record; z read twice as in the real kernel when TWICE
template <int TWICE>
__global__ void __launch_bounds__(BLOCK) k_strided(const u64* __restrict__ in, u64* __restrict__ out, uint8_t* __restrict__ inf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const u64x2* p = reinterpret_cast<const u64x2*>(in + 18 * i);
        u64x2 v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = p[k];
        if (TWICE) {
#pragma unroll
            for (int k = 6; k < 9; k++) { u64x2 w = __builtin_nontemporal_load(p + k); v[k].x ^= w.y; }
        }
        u64x2* q = reinterpret_cast<u64x2*>(out + 12 * i);
#pragma unroll
        for (int k = 0; k < 6; k++) { u64x2 w; w.x = v[k].x + v[(k + 3) % 9].y; w.y = v[k].y ^ v[(k + 6) % 9].x; q[k] = w; }
        inf[i] = (uint8_t)(v[8].x & 1);
    }
}
// (b) wave-contiguous: a workgroup moves 256 records per step as 9 (in) / 6 (out) contiguous 4 KiB spans
__global__ void __launch_bounds__(BLOCK) k_coalesced(const u64* __restrict__ in, u64* __restrict__ out, uint8_t* __restrict__ inf, size_t n) {
    const size_t ntiles = n / BLOCK;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u64x2* p = reinterpret_cast<const u64x2*>(in + 18 * tile * BLOCK);
        u64x2 v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = p[k * BLOCK + threadIdx.x];
        u64x2* q = reinterpret_cast<u64x2*>(out + 12 * tile * BLOCK);
#pragma unroll
        for (int k = 0; k < 6; k++) { u64x2 w; w.x = v[k].x + v[(k + 3) % 9].y; w.y = v[k].y ^ v[(k + 6) % 9].x; q[k * BLOCK + threadIdx.x] = w; }
        inf[tile * BLOCK + threadIdx.x] = (uint8_t)(v[8].x & 1);
    }
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 24;
    const size_t n = (size_t)1 << lg;
    u64 *in, *out; uint8_t* inf;
    CHECK(hipMalloc(&in, n * 144)); CHECK(hipMalloc(&out, n * 96)); CHECK(hipMalloc(&inf, n));
    CHECK(hipMemset(in, 1, n * 144));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int grid_mult : {2, 4, 8, 16}) {
        const int grid = 256 * grid_mult;
        for (int variant = 0; variant < 3; variant++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipEventRecord(e0));
                if (variant == 0) hipLaunchKernelGGL(k_strided<0>, dim3(grid), dim3(BLOCK), 0, 0, in, out, inf, n);
                else if (variant == 1) hipLaunchKernelGGL(k_strided<1>, dim3(grid), dim3(BLOCK), 0, 0, in, out, inf, n);
                else hipLaunchKernelGGL(k_coalesced, dim3(grid), dim3(BLOCK), 0, 0, in, out, inf, n);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("2^%d records, %2d WG/CU, %-22s %.3f ms  %.2f G rec/s  %.0f GB/s (241 B)\n", lg, grid_mult,
                   variant == 0 ? "strided" : variant == 1 ? "strided, z twice" : "wave-contiguous", best, n / best / 1e6, n * 241.0 / best / 1e6);
        }
    }
    return 0;
}
