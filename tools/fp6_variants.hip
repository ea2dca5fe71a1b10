// Measured alternatives to the Acc3 form of Fp6::mul: one wave-Fp6::mul timed in a dependent
// chain, occupancy chosen on the command line.   usage: fp6_variants [waves_per_simd = 4]
//
//   V0  acc3        the product path's fp6_mul_inl: 2 x 32-bit limbs both sides, 4 v_mad_u64_u32 + 4 v_addc_co_u32 per
//                   64 x 64 product in three columns with carry counters, one fold per coefficient (fp.hip.h)
//   V1  carryfree   asymmetric limbs: a in 2 x 32 bits, b and 7 b in 3 limbs of 22 / 22 / 20 (25 / 25 / 23) bits, so a
//                   coefficient's six 54..57-bit partial products add up below 2^60 in each of SIX 64-bit columns with no
//                   carry at all: 36 mads per coefficient and no addc.  The six columns (weights 0, 22, 32, 44, 54, 76)
//                   are brought to the three aligned columns + small top word of Acc3 by 7 more mads with constant
//                   2^22 / 2^12 multipliers, then the same fold.  Bit-identical canonical results (checked below).
//   V2  hiadd       "hi-word sums on full-rate v_add_u32": each limb product as D = a b + {lo, 0} (cannot carry), the
//                   high word summed by v_add_u32, then zeroed for the next mad's addend (v_mov).  Only the issue cost
//                   of that pattern is measured: the 32-bit high-word sums overflow for 32-bit limbs (six words of up
//                   to 2^32 - 1), so the form is not a correct multiplication without 29-bit limbs, i.e. without V1's
//                   extra mads -- a lower bound on what such a form could cost.
//   V3  f64fma      22-bit limbs in doubles, exact limb products by the two-FMA split (hi = a b rounded, lo = fma(a, b,
//                   -hi)), sums in doubles: 3 x 3 limb products x (2 fma + 2 add) per 64 x 64 product.  Issue cost of
//                   the pattern only (the carry propagation back to integers is not even counted).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../cheetah_amd/csrc/fp6.hip.h"
using namespace cheetah;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

#ifndef ITER
#define ITER 1000
#endif

// ---- V1 ---------------------------------------------------------------------------------------------------
struct B3 { u32 l[3]; };
CH_DEV B3 split22(u64 b) {
    B3 r;
    const u32 lo = lo32(b), hi = hi32(b);
    r.l[0] = lo & 0x3fffffu;
    r.l[1] = ((lo >> 22) | (hi << 10)) & 0x3fffffu;
    r.l[2] = hi >> 12;
    return r;
}
CH_DEV Fp6 fp6_mul_carryfree(const Fp6& a, const Fp6& b) {
    B3 bb[6], b7[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        bb[j] = split22(b.c[j]);
#pragma unroll
        for (int t = 0; t < 3; t++) b7[j].l[t] = bb[j].l[t] * 7u;  // limb-wise: no carry between limbs is needed
    }
    Fp6 r;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        u64 P0[3] = {0, 0, 0}, P1[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const B3& B = (i <= k) ? bb[k - i] : b7[k + 6 - i];
            const u32 al = lo32(a.c[i]), ah = hi32(a.c[i]);
#pragma unroll
            for (int t = 0; t < 3; t++) {
                P0[t] += (u64)al * B.l[t];
                P1[t] += (u64)ah * B.l[t];
            }
        }
        // V = P0[0] + 2^22 P0[1] + 2^44 P0[2] + 2^32 (P1[0] + 2^22 P1[1] + 2^44 P1[2]); every column < 2^60
        Acc3 acc;
        acc.c0 = P0[0] + ((u64)lo32(P0[1]) << 22);
        acc.c1 = P1[0] + ((u64)hi32(P0[1]) << 22) + ((u64)lo32(P0[2]) << 12) + ((u64)lo32(P1[1]) << 22);
        acc.c2 = ((u64)hi32(P0[2]) << 12) + ((u64)hi32(P1[1]) << 22) + ((u64)lo32(P1[2]) << 12);
        const u64 w3 = (u64)hi32(P1[2]) << 12;  // weight 2^96
        acc.t0 = 0;
        acc.t1 = lo32(w3);
        acc.t2 = hi32(w3);
        r.c[k] = acc.reduce();
    }
    return r;
}

// ---- V2: issue pattern only -------------------------------------------------------------------------------------
CH_DEV Fp6 fp6_mul_hiadd_pattern(const Fp6& a, const Fp6& b) {
    u64 b7[6];
    b7[0] = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) b7[j] = fp_mul_small(b.c[j], 7u);
    Fp6 r;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        u32 l0 = 0, l1 = 0, l2 = 0, h0 = 0, h1 = 0, h2 = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const u64 y = (i <= k) ? b.c[k - i] : b7[k + 6 - i];
            const u32 al = lo32(a.c[i]), ah = hi32(a.c[i]), bl = lo32(y), bh = hi32(y);
            u64 d0 = l0, d1 = l1, d2 = l2, d3;  // {lo, 0}
            asm("v_mad_u64_u32 %[d0], vcc, %[al], %[bl], %[d0]\n\t"
                "v_mad_u64_u32 %[d1], vcc, %[al], %[bh], %[d1]\n\t"
                "v_mad_u64_u32 %[d2], vcc, %[ah], %[bh], %[d2]\n\t"
                : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2)
                : [al] "v"(al), [ah] "v"(ah), [bl] "v"(bl), [bh] "v"(bh)
                : "vcc");
            h0 += hi32(d0);
            h1 += hi32(d1);
            h2 += hi32(d2);
            d3 = lo32(d1);
            asm("v_mad_u64_u32 %[d3], vcc, %[ah], %[bl], %[d3]" : [d3] "+v"(d3) : [ah] "v"(ah), [bl] "v"(bl) : "vcc");
            h1 += hi32(d3);
            l0 = lo32(d0);
            l1 = lo32(d3);
            l2 = lo32(d2);
        }
        Acc3 acc;
        acc.c0 = mk64(l0, h0);
        acc.c1 = mk64(l1, h1);
        acc.c2 = mk64(l2, h2);
        acc.t0 = acc.t1 = acc.t2 = 0;
        r.c[k] = acc.reduce();
    }
    return r;
}

// ---- V3: issue pattern only -------------------------------------------------------------------------------------
CH_DEV Fp6 fp6_mul_f64_pattern(const Fp6& a, const Fp6& b) {
    double A[6][3], B[6][3];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const B3 x = split22(a.c[j]), y = split22(b.c[j]);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            A[j][t] = (double)x.l[t];
            B[j][t] = (double)y.l[t];
        }
    }
    Fp6 r;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double hi[5] = {0, 0, 0, 0, 0}, lo[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int j = (k - i + 6) % 6;
            const double w = (i <= k) ? 1.0 : 7.0;
#pragma unroll
            for (int s = 0; s < 3; s++) {
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const double y = B[j][t] * w;
                    const double h = A[i][s] * y;
                    const double l = __builtin_fma(A[i][s], y, -h);
                    hi[s + t] += h;
                    lo[s + t] += l;
                }
            }
        }
        u64 v = 0;
#pragma unroll
        for (int c = 0; c < 5; c++) v += (u64)(long long)hi[c] + (u64)(long long)lo[c];
        r.c[k] = v;
    }
    return r;
}

template <int V>
__global__ void __launch_bounds__(256) k_chain(const u64* in, u64* out, int iters) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp6 x, y;
    for (int l = 0; l < 6; l++) { x.c[l] = in[l] + i * 0x9E3779B97F4A7C15ull; y.c[l] = in[6 + l] ^ (i * 0xBF58476D1CE4E5B9ull); }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (V == 0) x = fp6_mul_inl(x, y);
        if (V == 1) x = fp6_mul_carryfree(x, y);
        if (V == 2) x = fp6_mul_hiadd_pattern(x, y);
        if (V == 3) x = fp6_mul_f64_pattern(x, y);
    }
    for (int l = 0; l < 6; l++) out[6 * i + l] = fp_canonical(x.c[l]);
}

typedef void (*kern_t)(const u64*, u64*, int);
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * waves;
    const size_t n = (size_t)blocks * 256;
    u64 *in, *o0, *o1; CHECK(hipMalloc(&in, 96)); CHECK(hipMalloc(&o0, n * 48)); CHECK(hipMalloc(&o1, n * 48));
    u64 h[12]; for (int i = 0; i < 12; i++) h[i] = 0xf23456789abcdef1ull * (i + 3) + 0xffffffff00000000ull * (i & 1);
    CHECK(hipMemcpy(in, h, 96, hipMemcpyHostToDevice));
    // V1 must equal V0 after a chain of 64 products on every lane
    hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, in, o0, 64);
    hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, in, o1, 64);
    CHECK(hipDeviceSynchronize());
    u64* a = (u64*)malloc(n * 48); u64* b = (u64*)malloc(n * 48);
    CHECK(hipMemcpy(a, o0, n * 48, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b, o1, n * 48, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < n * 6; i++) bad += a[i] != b[i];
    struct { const char* name; kern_t f; } ks[] = {{"acc3", k_chain<0>}, {"carryfree", k_chain<1>}, {"hiadd_pattern", k_chain<2>}, {"f64fma_pattern", k_chain<3>}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"waves_per_simd\": %d, \"carryfree_equals_acc3\": %s", waves, bad ? "false" : "true");
    for (auto& e : ks) {
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, in, o0, ITER); CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        // one SIMD runs `waves` waves: SIMD-ns per wave-product = time / (iters * waves)
        printf(", \"%s\": {\"ns_per_wave_product\": %.1f, \"G_products_per_s\": %.2f}", e.name, best * 1e6 / ITER / waves, (double)n * ITER / (best * 1e-3) / 1e9);
    }
    printf("}\n");
    return bad ? 1 : 0;
}
