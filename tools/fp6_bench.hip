// Micro-benchmark of the Fp6 device functions in isolation: each lane runs a
// dependent chain of ITER operations; reports ns per op per lane-wave and the
// implied chip rate.  usage: fp6_bench [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../cheetah_amd/csrc/curve.hip.h"
using namespace cheetah;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
constexpr int ITER = 2000;

template <int OP>
__global__ void __launch_bounds__(256) k(const u64* in, u64* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp6 x, y;
    for (int l = 0; l < 6; l++) { x.c[l] = in[l] + i; y.c[l] = in[6 + l] ^ i; }
    if (OP == 0) for (int it = 0; it < ITER; it++) x = fp6_mul(x, y);
    if (OP == 1) for (int it = 0; it < ITER; it++) x = fp6_square(x);
    if (OP == 2) for (int it = 0; it < ITER; it++) x = fp6_add(x, y);
    if (OP == 3) for (int it = 0; it < ITER; it++) x = fp6_sub(x, y);
    if (OP == 4) { Proj p; p.x = x; p.y = y; p.z = fp6_add(x, y); for (int it = 0; it < ITER / 10; it++) p = proj_double_unchecked(p); x = fp6_add(p.x, fp6_add(p.y, p.z)); }
    if (OP == 5) { Proj p; p.x = x; p.y = y; p.z = fp6_add(x, y); Affine q; q.x = y; q.y = x; q.inf = false; for (int it = 0; it < ITER / 10; it++) p = proj_add_mixed_unchecked(p, q); x = fp6_add(p.x, fp6_add(p.y, p.z)); }
    if (OP == 6 || OP == 8) { MJac p; p.x = x; p.y = y; p.z = fp6_add(x, y); p.z4 = fp6_sub(x, y); for (int it = 0; it < ITER / 10; it++) p = OP == 6 ? mjac_double_unchecked(p) : mjac_double_unchecked_stale_z4(p); x = fp6_add(fp6_add(p.x, p.z4), fp6_add(p.y, p.z)); }
    if (OP == 7 || OP == 9) { MJac p; p.x = x; p.y = y; p.z = fp6_add(x, y); p.z4 = fp6_sub(x, y); Affine q; q.x = y; q.y = x; q.inf = false; for (int it = 0; it < ITER / 10; it++) p = OP == 7 ? mjac_add_mixed_unchecked(p, q) : mjac_add_mixed_unchecked_fresh_z4<true>(p, q); x = fp6_add(fp6_add(p.x, p.z4), fp6_add(p.y, p.z)); }
    if (OP == 10) for (int it = 0; it < ITER; it++) x = fp6_double(x);
    if (OP == 11) for (int it = 0; it < ITER; it++) x = fp6_mul_small(x, 4u);
    if (OP == 12) for (int it = 0; it < ITER; it++) x = fp6_triple(x);
    for (int l = 0; l < 6; l++) out[6 * i + l] = x.c[l];
}
typedef void (*kern_t)(const u64*, u64*);
int main(int argc, char** argv) {
    int bpc = argc > 1 ? atoi(argv[1]) : 2;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * bpc;
    u64 *in, *out; CHECK(hipMalloc(&in, 96)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 48));
    u64 h[12]; for (int i = 0; i < 12; i++) h[i] = 0x123456789abcdefull * (i + 3);
    CHECK(hipMemcpy(in, h, 96, hipMemcpyHostToDevice));
    struct { const char* name; kern_t f; int iters; } ks[] = {{"fp6_mul", k<0>, ITER}, {"fp6_square", k<1>, ITER}, {"fp6_add", k<2>, ITER}, {"fp6_sub", k<3>, ITER},
        {"proj_double_unchecked", k<4>, ITER / 10}, {"proj_add_mixed_unchecked", k<5>, ITER / 10},
        {"mjac_double_unchecked", k<6>, ITER / 10}, {"mjac_add_mixed_unchecked", k<7>, ITER / 10},
        {"mjac_double_stale_z4", k<8>, ITER / 10}, {"mjac_add_fresh_z4", k<9>, ITER / 10},
        {"fp6_double", k<10>, ITER}, {"fp6_mul_small4", k<11>, ITER}, {"fp6_triple", k<12>, ITER}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"blocks_per_cu\": %d", bpc);
    for (auto& e : ks) {
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, in, out); CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double ops = (double)blocks * 256 * e.iters;
        // cycles per op per wave at 2.4 GHz: each SIMD runs bpc waves
        double cyc = best * 1e-3 * 2.4e9 / e.iters / bpc;
        printf(", \"%s\": {\"Gops_s\": %.2f, \"simd_cycles_per_wave_op\": %.0f}", e.name, ops / (best * 1e-3) / 1e9, cyc);
    }
    printf("}\n");
    return 0;
}
