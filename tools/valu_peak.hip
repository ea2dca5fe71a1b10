// Integer-VALU peak calibration for gfx950: issue rate of the instructions the
// Goldilocks arithmetic is made of.  Each kernel runs ILP independent chains of
// one instruction per lane; rate = lanes * ops / time.  Prints one JSON object.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int ILP = 8;
constexpr int ITERS = 8192;  // loop trips
constexpr int UNROLL = 8;    // ops per chain per trip

#define DEF_KERNEL32(NAME, ASMSTR)                                                        \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {               \
    unsigned a[ILP], b = threadIdx.x * 2654435761u + seed, c = b ^ 0x9e3779b9u;          \
    for (int i = 0; i < ILP; i++) a[i] = b + i * 77u;                                     \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
    for (int it = 0; it < ITERS; it++) {                                                  \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                              \
            _Pragma("unroll") for (int i = 0; i < ILP; i++)                               \
                asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c));                       \
        }                                                                                 \
    }                                                                                     \
    unsigned r = 0; for (int i = 0; i < ILP; i++) r ^= a[i];                              \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(t1 - t0) + (r == 0x12345u);   \
}

DEF_KERNEL32(k_add_u32,      "v_add_u32 %0, %0, %1")
DEF_KERNEL32(k_add_co_u32,   "v_add_co_u32 %0, vcc, %0, %1")
DEF_KERNEL32(k_addc_co_u32,  "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF_KERNEL32(k_mul_lo_u32,   "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL32(k_mul_hi_u32,   "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL32(k_mad_u32_u24,  "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL32(k_mul_u32_u24,  "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL32(k_add3_u32,     "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL32(k_cndmask,      "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL32(k_alignbit,     "v_alignbit_b32 %0, %0, %1, 29")
DEF_KERNEL32(k_lshl_or,      "v_lshl_or_b32 %0, %0, 3, %1")
DEF_KERNEL32(k_xor,          "v_xor_b32 %0, %0, %1")
DEF_KERNEL32(k_mad_u32_u16,  "v_mad_u32_u16 %0, %0, %1, %2")
DEF_KERNEL32(k_dot2_u32_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
DEF_KERNEL32(k_dot4_u32_u8,  "v_dot4_u32_u8 %0, %1, %2, %0")
DEF_KERNEL32(k_sub_co_u32,   "v_sub_co_u32 %0, vcc, %0, %1")
DEF_KERNEL32(k_subb_co_u32,  "v_subb_co_u32 %0, vcc, %0, %1, vcc")

#define DEF_KERNEL64(NAME, ASMSTR)                                                        \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {               \
    unsigned long long a[ILP];                                                            \
    unsigned b = threadIdx.x * 2654435761u + seed, c = b ^ 0x9e3779b9u;                   \
    unsigned long long bb = ((unsigned long long)b << 32) | c;                            \
    for (int i = 0; i < ILP; i++) a[i] = bb + i * 77u;                                    \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
    for (int it = 0; it < ITERS; it++) {                                                  \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                              \
            _Pragma("unroll") for (int i = 0; i < ILP; i++)                               \
                asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c), "v"(bb));              \
        }                                                                                 \
    }                                                                                     \
    unsigned long long r = 0; for (int i = 0; i < ILP; i++) r ^= a[i];                    \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(t1 - t0) + (r == 0x12345u);   \
}

DEF_KERNEL64(k_mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
DEF_KERNEL64(k_fma_f64,      "v_fma_f64 %0, %3, %3, %0")
DEF_KERNEL64(k_add_f64,      "v_add_f64 %0, %0, %3")
DEF_KERNEL64(k_lshlrev_b64,  "v_lshlrev_b64 %0, 3, %0")

typedef void (*kern_t)(unsigned*, unsigned);

static double g_cyc = 0;
static double run(kern_t k, int blocks, unsigned* d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, (unsigned)r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned> h((size_t)blocks * 256);
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::vector<unsigned> w; for (size_t i = 0; i < h.size(); i += 64) w.push_back(h[i]);
    std::sort(w.begin(), w.end());
    g_cyc = (double)w[w.size() / 2] / ((double)ILP * ITERS * UNROLL);
    double ops = (double)blocks * 256.0 * ILP * ITERS * UNROLL;
    return ops / (best * 1e-3) / 1e12;  // T lane-ops / s
}

int main(int argc, char** argv) {
    int wavesPerSimd = argc > 1 ? atoi(argv[1]) : 2;   // resident waves per SIMD
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * wavesPerSimd;      // 256-thread block = 4 waves = 1 per SIMD; all resident at once
    unsigned* d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    struct { const char* name; kern_t k; } ks[] = {
        {"v_add_u32", k_add_u32}, {"v_add_co_u32", k_add_co_u32}, {"v_addc_co_u32", k_addc_co_u32},
        {"v_sub_co_u32", k_sub_co_u32}, {"v_subb_co_u32", k_subb_co_u32},
        {"v_add3_u32", k_add3_u32}, {"v_xor_b32", k_xor}, {"v_cndmask_b32", k_cndmask}, {"v_alignbit_b32", k_alignbit},
        {"v_lshl_or_b32", k_lshl_or},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mad_u32_u16", k_mad_u32_u16}, {"v_dot2_u32_u16", k_dot2_u32_u16},
        {"v_dot4_u32_u8", k_dot4_u32_u8},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshlrev_b64", k_lshlrev_b64},
        {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64},
    };
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"grid_blocks\": %d, \"waves_per_simd\": %d, \"T_lane_ops_per_s__cycles_per_wave_instr\": {", prop.name, cus, prop.clockRate / 1000, blocks, wavesPerSimd);
    bool first = true;
    for (auto& e : ks) {
        double r = run(e.k, blocks, d);
        printf("%s\"%s\": [%.3f, %.2f]", first ? "" : ", ", e.name, r, g_cyc);
        first = false;
        fflush(stdout);
    }
    printf("}}\n");
    return 0;
}
