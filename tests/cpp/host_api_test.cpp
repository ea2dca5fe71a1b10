// Exercises the C++ host API (cheetah_amd/host/cheetah.hpp) against the
// reference's known-answer vectors.  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstring>

#include "../../cheetah_amd/host/cheetah.hpp"

using namespace cheetah;

static int fails = 0;
#define EXPECT(cond)                                                  \
    do {                                                              \
        if (!(cond)) {                                                \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); \
            fails++;                                                  \
        }                                                             \
    } while (0)

int main() {
    try {
        Context ctx(0);
        // Fp6 mul KAT, src/fp6.rs:1602-1627
        Fp6 a{{7311251207099623982ull, 11101156575191577169ull, 3471450285948349450ull, 18081717484378824777ull,
               17271827452459074620ull, 13245881484175979175ull}};
        Fp6 b{{10376570592177434053ull, 2266730037751779421ull, 17604547680137376634ull, 8522386576290610241ull,
               4622828580943454349ull, 3310945140500303981ull}};
        Fp6 c{{16699204345089849089ull, 12290068205868725576ull, 15503892176250492698ull, 16153850030910369456ull,
               4357822170351527454ull, 15609745236686299732ull}};
        auto r = fp6_mul_batch(ctx, {a}, {b});
        EXPECT(r[0].c == c.c);
        // Fp6 square KAT, src/fp6.rs:1532-1551
        Fp6 sq{{18442855563257030637ull, 416354386888665317ull, 12488382742185552506ull, 6173432951622501799ull,
                9922025629450424125ull, 9218000164529386038ull}};
        EXPECT(fp6_square_batch(ctx, {a})[0].c == sq.c);
        // invert(0) is none, invert(1) = 1, src/fp6.rs:1754-1756
        std::vector<uint8_t> some;
        Fp6 one{{1, 0, 0, 0, 0, 0}}, zero{{0, 0, 0, 0, 0, 0}};
        auto inv = fp6_invert_batch(ctx, {one, zero}, some);
        EXPECT(inv[0].c == one.c && some[0] == 1 && some[1] == 0 && inv[1].c == zero.c);
        // 2 * G in affine, src/curve/projective.rs:1157-1178
        Scalar two{};
        two.bytes[0] = 2;
        std::vector<uint8_t> infinity;
        auto g2 = BASEPOINT_TABLE.multiply_batch_affine(ctx, {two}, infinity);
        Fp6 x2{{0x7f4c1bfc52278ad8ull, 0xfa8e921f7580e371ull, 0x97252bf35d1c7668ull, 0xe6d0901604cae95aull,
                0xae36bba2ad2ee0d7ull, 0x194b4e35a2a9c77ull}};
        Fp6 y2{{0x144045efbce03ef8ull, 0x8e5fe3f66f8b370dull, 0x3d54df63b96bfd20ull, 0x2418219e37948caaull,
                0xd4c1a40432582552ull, 0x367b029f5f146e3dull}};
        EXPECT(g2[0].x.c == x2.c && g2[0].y.c == y2.c && infinity[0] == 0);
        // projective result of the comb normalises to the same point; identity comes out as (0, 1, inf)
        auto gp = BASEPOINT_TABLE.multiply_batch(ctx, {two});
        std::vector<ProjectivePoint> pts = {gp[0], ProjectivePoint{zero, one, zero}};
        std::vector<AffinePoint> q;
        batch_normalize(ctx, pts, q, infinity);
        EXPECT(q[0].x.c == x2.c && q[0].y.c == y2.c && infinity[0] == 0);
        EXPECT(q[1].x.c == zero.c && q[1].y.c == one.c && infinity[1] == 1);
        // variable base: 2 * (2G) == 4 * G by both ladders; q * G = O (is_torsion_free, src/curve/projective.rs:885-893)
        Scalar four{};
        four.bytes[0] = 4;
        auto g4 = BASEPOINT_TABLE.multiply_batch_affine(ctx, {four}, infinity);
        auto m = multiply_batch(ctx, std::vector<AffinePoint>{g2[0]}, {two}, infinity);
        EXPECT(m[0].x.c == g4[0].x.c && m[0].y.c == g4[0].y.c);
        const uint8_t qb[32] = {207, 172, 212, 174, 62, 98, 67, 212, 34, 119, 21, 48, 35, 167, 122, 50,
                                181, 55, 10, 153, 15, 191, 63, 86, 208, 34, 63, 59, 155, 89, 242, 122};
        Scalar qs{};
        std::memcpy(qs.bytes.data(), qb, 32);
        auto o = multiply_vartime_batch(ctx, {gp[0]}, {qs});
        EXPECT(o[0].z.c == zero.c);
        // is_torsion_free(G) (src/curve/projective.rs:1462); 1*(2G) + 1*(2G) == 4*G through multiply_double;
        // 1*(2G) + 2*G == 4*G through the basepoint ladder; BASEPOINT_LOOKUP * 2 normalises to 2G
        EXPECT(is_torsion_free_batch(ctx, {gp[0]})[0] == 1);
        Scalar one_s{};
        one_s.bytes[0] = 1;
        auto dd = multiply_many_batch(ctx, {gp[0], gp[0]}, {one_s, one_s}, 2);
        auto db = multiply_double_with_basepoint_vartime_batch(ctx, {gp[0]}, {one_s}, {two});
        auto jl = BASEPOINT_LOOKUP.multiply_batch(ctx, {two});
        std::vector<ProjectivePoint> chk = {dd[0], db[0]};
        batch_normalize(ctx, chk, q, infinity);
        EXPECT(q[0].x.c == g4[0].x.c && q[0].y.c == g4[0].y.c && q[1].x.c == g4[0].x.c && q[1].y.c == g4[0].y.c);
        std::vector<JacobianPoint> jchk = {jl[0]};
        batch_normalize(ctx, jchk, q, infinity);
        EXPECT(q[0].x.c == x2.c && q[0].y.c == y2.c);
        // encodings: identity byte strings (src/curve/projective.rs:1533-1539, :1575-1582), 2G -> 49 bytes -> 2G,
        // sqrt(1) = 1 and u + 2 is a non-square (src/fp6.rs:1561-1577), from_x(1) is none (src/curve/affine.rs:645-653)
        {
            std::vector<AffinePoint> ap = {g2[0], AffinePoint{zero, one}};
            std::vector<uint8_t> ainf = {0, 1}, dinf, dsome;
            auto cp = to_compressed_batch(ctx, ap, ainf);
            auto up = to_uncompressed_batch(ctx, ap, ainf);
            bool id_ok = cp[1].bytes[48] == 128 && up[1].bytes[96] == 128;
            for (int i = 0; i < 48; i++) id_ok = id_ok && cp[1].bytes[i] == 0;
            for (int i = 0; i < 96; i++) id_ok = id_ok && up[1].bytes[i] == 0;
            EXPECT(id_ok);
            auto dp = from_compressed_batch(ctx, cp, dinf, dsome, true);
            EXPECT(dsome[0] == 1 && dsome[1] == 1 && dp[0].x.c == x2.c && dp[0].y.c == y2.c && dinf[0] == 0 && dinf[1] == 1);
            auto du = from_uncompressed_batch(ctx, up, dinf, dsome, true);
            EXPECT(dsome[0] == 1 && dsome[1] == 1 && du[0].x.c == x2.c && du[0].y.c == y2.c && dinf[1] == 1);
            Fp6 nsq{{2, 1, 0, 0, 0, 0}};
            auto rt = fp6_sqrt_batch(ctx, {one, nsq, zero}, dsome);
            EXPECT(rt[0].c == one.c && dsome[0] == 1 && dsome[1] == 0 && rt[1].c == zero.c && dsome[2] == 1 && rt[2].c == zero.c);
            auto fx = from_x_batch(ctx, {one, zero, x2}, dinf, dsome);
            EXPECT(dsome[0] == 0 && dsome[1] == 1 && dinf[1] == 1 && dsome[2] == 1 && fx[2].x.c == x2.c);
        }
        // group law over batches (src/curve/projective.rs tests test_projective_addition / test_doubling / negation):
        // 2G + 2G == double(2G) == 4G, 2G + affine 2G too, 2G + (-2G) is the identity, double_multi(2)(G-multiple) == 4x
        {
            std::vector<ProjectivePoint> p2 = {gp[0]};
            auto s1 = add_batch(ctx, p2, p2);
            auto s2 = add_mixed_batch(ctx, p2, std::vector<AffinePoint>{g2[0]});
            auto d1 = double_batch(ctx, p2);
            auto n1 = neg_batch(ctx, p2);
            auto z1 = add_batch(ctx, p2, n1);
            auto d2 = double_multi_batch(ctx, p2, 2);                  // 8G
            auto g8 = add_batch(ctx, d1, d1);
            EXPECT(eq_batch(ctx, s1, d1)[0] == 1 && eq_batch(ctx, s2, d1)[0] == 1 && eq_batch(ctx, d2, g8)[0] == 1);
            EXPECT(eq_batch(ctx, s1, p2)[0] == 0 && z1[0].z.c == zero.c);
            std::vector<ProjectivePoint> chk4 = {s1[0], d1[0]};
            batch_normalize(ctx, chk4, q, infinity);
            EXPECT(q[0].x.c == g4[0].x.c && q[0].y.c == g4[0].y.c && q[1].x.c == g4[0].x.c && q[1].y.c == g4[0].y.c);
        }
        // round 3's widening: Fp::sqrt KATs (src/fp.rs:899-910: sqrt(0) = 0, sqrt(1) = 1, 7 is not a square), Fp::exp
        // (a^(p-2) a = 1, :914-934), Fp6::exp (a^0 = 1, a^1 = a, a^2 = a.square(); the long exponents are compared with
        // the oracle in tests/test_gpu_widening.py), seeded random points (on the curve's prime-order subgroup, the affine
        // and the projective form of one stream the same points, a window of a stream the same elements)
        {
            const uint64_t P = 0xFFFFFFFF00000001ull;
            std::vector<uint8_t> some;
            auto r = fp_sqrt_batch(ctx, {0, 1, 7, 4, P + 1}, some);
            EXPECT(some[0] == 1 && r[0] == 0 && some[1] == 1 && r[1] == 1 && some[2] == 0 && some[3] == 1 && (r[3] == 2 || r[3] == P - 2));
            EXPECT(some[4] == 1 && r[4] == 1);
            auto e = fp_exp_batch(ctx, {3, 5, P - 1}, P - 2);
            auto back = fp6_mul_batch(ctx, {Fp6{{e[0], 0, 0, 0, 0, 0}}, Fp6{{e[1], 0, 0, 0, 0, 0}}}, {Fp6{{3, 0, 0, 0, 0, 0}}, Fp6{{5, 0, 0, 0, 0, 0}}});
            EXPECT(back[0].c == one.c && back[1].c == one.c && e[2] == P - 1);
            auto p1 = fp6_exp_batch(ctx, {a, b}, {1, 0, 0, 0, 0, 0});
            auto p0 = fp6_exp_batch(ctx, {a, b}, {0, 0, 0, 0, 0, 0});
            auto p2 = fp6_exp_batch(ctx, {a, b}, {2, 0, 0, 0, 0, 0});
            auto sq = fp6_square_batch(ctx, {a, b});
            EXPECT(p1[0].c == fp6_mul_batch(ctx, {a}, {one})[0].c && p0[0].c == one.c && p0[1].c == one.c && p2[0].c == sq[0].c && p2[1].c == sq[1].c);
            std::vector<uint8_t> rinf;
            auto ra = random_affine_batch(ctx, 77, 300, rinf);
            auto rp = random_projective_batch(ctx, 77, 300);
            std::vector<AffinePoint> rq;
            std::vector<uint8_t> rqinf;
            batch_normalize(ctx, rp, rq, rqinf);
            auto tf = is_torsion_free_batch(ctx, rp);
            bool same = true, free_ = true, none_inf = true;
            for (size_t i = 0; i < ra.size(); i++) {
                same = same && ra[i].x.c == rq[i].x.c && ra[i].y.c == rq[i].y.c;
                free_ = free_ && tf[i] == 1;
                none_inf = none_inf && rinf[i] == 0 && rqinf[i] == 0;
            }
            EXPECT(same && free_ && none_inf);
            auto ra2 = random_affine_batch(ctx, 77, 10, rinf, 100);          // a later window of the same stream
            EXPECT(ra2[0].x.c == ra[100].x.c && ra2[9].y.c == ra[109].y.c);
        }
        // length mismatch is an error, not UB
        bool threw = false;
        try {
            fp6_mul_batch(ctx, {a, a}, {b});
        } catch (const Error&) {
            threw = true;
        }
        EXPECT(threw);
    } catch (const Error& e) {
        std::printf("cheetah::Error: %s\n", e.what());
        return 2;
    }
    std::printf(fails ? "host_api_test: %d FAILED\n" : "host_api_test: ok\n", fails);
    return fails ? 1 : 0;
}
