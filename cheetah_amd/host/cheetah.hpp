// cheetah.hpp -- C++ host-side mirror of the reference crate's API for the batched
// hot path, on top of the C ABI (include/cheetah_hip.h).
//
// The reference is Rust (toposware/cheetah); no Rust toolchain exists in this
// pipeline, so the host side above the C ABI is written in C++ with the
// reference's names and argument meaning: cheetah::Fp6, cheetah::AffinePoint,
// cheetah::ProjectivePoint, cheetah::JacobianPoint, cheetah::BASEPOINT_TABLE.
// Each batch method is the slice form of the reference method of the same name
// (file:line cited at the method).  Errors surface as cheetah::Error, mirroring
// the reference's panics on length mismatch (e.g. batch_normalize's
// assert_eq!(p.len(), q.len()), src/curve/projective.rs:898).
//
// Header-only; link with -lcheetah_hip.  No arithmetic happens on the host.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cheetah_hip.h"

namespace cheetah {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("cheetah_hip error " + std::to_string(c) + ": " + m), code(c) {}
};

// Plain-old-data images of the reference types, in the layout of the C ABI.
struct Fp6 {
    std::array<uint64_t, 6> c;  // Fp6::output_internal(), src/fp6.rs:1126-1135
};
struct Scalar {
    std::array<uint8_t, 32> bytes;  // Scalar::to_bytes(), src/scalar.rs:477-489
};
struct AffinePoint {
    Fp6 x, y;  // infinity flags travel in a separate array (src/curve/affine.rs:37-41)
};
struct ProjectivePoint {
    Fp6 x, y, z;  // src/curve/projective.rs:76-80
};
struct JacobianPoint {
    Fp6 x, y, z;  // src/curve/jacobian.rs:76-80
};
static_assert(sizeof(Fp6) == 48 && sizeof(AffinePoint) == 96 && sizeof(ProjectivePoint) == 144, "ABI layout");

// One GPU or several.  Owns the device tables the reference builds with lazy_static
// (BASEPOINT_TABLE, shift points; src/constants.rs:18-90), replicated per device; with several
// devices every batch call is sharded contiguously over them and gathered into the caller's arrays.
class Context {
public:
    explicit Context(int device = 0) {
        int rc = cheetah_ctx_create(&device, 1, &h_);
        if (rc != 0) throw Error(rc, cheetah_last_error(nullptr));
    }
    explicit Context(const std::vector<int>& devices) {
        int rc = cheetah_ctx_create(devices.data(), (int)devices.size(), &h_);
        if (rc != 0) throw Error(rc, cheetah_last_error(nullptr));
    }
    ~Context() {
        if (h_) cheetah_ctx_destroy(h_);
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    cheetah_ctx* get() const { return h_; }
    void check(int rc) const {
        if (rc != 0) throw Error(rc, cheetah_last_error(h_));
    }

private:
    cheetah_ctx* h_ = nullptr;
};

inline const uint64_t* u64p(const void* p) { return static_cast<const uint64_t*>(p); }
inline uint64_t* u64p(void* p) { return static_cast<uint64_t*>(p); }

// ---- Fp6 --------------------------------------------------------------------
// out[i] = a[i] * b[i]          Fp6::mul, src/fp6.rs:651-834
inline std::vector<Fp6> fp6_mul_batch(const Context& ctx, const std::vector<Fp6>& a, const std::vector<Fp6>& b) {
    if (a.size() != b.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<Fp6> out(a.size());
    ctx.check(cheetah_fp6_mul_batch(ctx.get(), u64p(a.data()), u64p(b.data()), u64p(out.data()), a.size()));
    return out;
}
// out[i] = a[i]^2               Fp6::square, src/fp6.rs:838-920
inline std::vector<Fp6> fp6_square_batch(const Context& ctx, const std::vector<Fp6>& a) {
    std::vector<Fp6> out(a.size());
    ctx.check(cheetah_fp6_square_batch(ctx.get(), u64p(a.data()), u64p(out.data()), a.size()));
    return out;
}
// out[i] = a[i]^-1, is_some[i]  Fp6::invert -> CtOption, src/fp6.rs:1035-1049
inline std::vector<Fp6> fp6_invert_batch(const Context& ctx, const std::vector<Fp6>& a, std::vector<uint8_t>& is_some) {
    std::vector<Fp6> out(a.size());
    is_some.assign(a.size(), 0);
    ctx.check(cheetah_fp6_invert_batch(ctx.get(), u64p(a.data()), u64p(out.data()), is_some.data(), a.size()));
    return out;
}
// out[i] = sqrt(a[i]), is_some[i]  Fp6::sqrt -> CtOption, src/fp6.rs:922-964
inline std::vector<Fp6> fp6_sqrt_batch(const Context& ctx, const std::vector<Fp6>& a, std::vector<uint8_t>& is_some) {
    std::vector<Fp6> out(a.size());
    is_some.assign(a.size(), 0);
    ctx.check(cheetah_fp6_sqrt_batch(ctx.get(), u64p(a.data()), u64p(out.data()), is_some.data(), a.size()));
    return out;
}

// out[i] = a[i]^power            Fp6::exp / exp_vartime, src/fp6.rs:1053-1086 (power: six little-endian u64 words)
inline std::vector<Fp6> fp6_exp_batch(const Context& ctx, const std::vector<Fp6>& a, const std::array<uint64_t, 6>& power) {
    std::vector<Fp6> out(a.size());
    ctx.check(cheetah_fp6_exp_batch(ctx.get(), u64p(a.data()), power.data(), u64p(out.data()), a.size()));
    return out;
}

// ---- Fp (values are any u64, results canonical; src/fp.rs) ---------------------------
// out[i] = sqrt(a[i]), is_some[i]  Fp::sqrt -> CtOption, src/fp.rs:199-229
inline std::vector<uint64_t> fp_sqrt_batch(const Context& ctx, const std::vector<uint64_t>& a, std::vector<uint8_t>& is_some) {
    std::vector<uint64_t> out(a.size());
    is_some.assign(a.size(), 0);
    ctx.check(cheetah_fp_op_batch(ctx.get(), CHEETAH_FP_SQRT, a.data(), nullptr, 0, out.data(), is_some.data(), a.size()));
    return out;
}
// out[i] = a[i]^power              Fp::exp / exp_vartime, src/fp.rs:304-333
inline std::vector<uint64_t> fp_exp_batch(const Context& ctx, const std::vector<uint64_t>& a, uint64_t power) {
    std::vector<uint64_t> out(a.size());
    ctx.check(cheetah_fp_op_batch(ctx.get(), CHEETAH_FP_EXP, a.data(), nullptr, power, out.data(), nullptr, a.size()));
    return out;
}

// ---- batch_normalize ----------------------------------------------------------
// ProjectivePoint::batch_normalize(p, q), src/curve/projective.rs:897-930
inline void batch_normalize(const Context& ctx, const std::vector<ProjectivePoint>& p, std::vector<AffinePoint>& q,
                            std::vector<uint8_t>& infinity) {
    q.resize(p.size());
    infinity.assign(p.size(), 0);
    ctx.check(cheetah_batch_normalize_projective(ctx.get(), u64p(p.data()), u64p(q.data()), infinity.data(), p.size()));
}
// JacobianPoint::batch_normalize(p, q), src/curve/jacobian.rs:886-920
inline void batch_normalize(const Context& ctx, const std::vector<JacobianPoint>& p, std::vector<AffinePoint>& q,
                            std::vector<uint8_t>& infinity) {
    q.resize(p.size());
    infinity.assign(p.size(), 0);
    ctx.check(cheetah_batch_normalize_jacobian(ctx.get(), u64p(p.data()), u64p(q.data()), infinity.data(), p.size()));
}

// ---- ladders --------------------------------------------------------------------
// &BASEPOINT_TABLE * &scalar, src/constants.rs:92-98 -> BasePointTable::multiply, src/lookup.rs:189-206
struct BasePointTableT {
    std::vector<ProjectivePoint> multiply_batch(const Context& ctx, const std::vector<Scalar>& by) const {
        std::vector<ProjectivePoint> out(by.size());
        ctx.check(cheetah_mul_by_generator_batch(ctx.get(), by.data()->bytes.data(), u64p(out.data()), nullptr, by.size(),
                                                 CHEETAH_PROJECTIVE));
        return out;
    }
    std::vector<AffinePoint> multiply_batch_affine(const Context& ctx, const std::vector<Scalar>& by,
                                                   std::vector<uint8_t>& infinity) const {
        std::vector<AffinePoint> out(by.size());
        infinity.assign(by.size(), 0);
        ctx.check(cheetah_mul_by_generator_batch(ctx.get(), by.data()->bytes.data(), u64p(out.data()), infinity.data(),
                                                 by.size(), CHEETAH_AFFINE));
        return out;
    }
};
static const BasePointTableT BASEPOINT_TABLE{};

// BasePointTable::create(&P) for a caller-supplied base, src/lookup.rs:164-173; multiply, :189-206
class BasePointTable {
public:
    BasePointTable(const Context& ctx, const ProjectivePoint& p) : ctx_(ctx) {
        ctx.check(cheetah_basepoint_table_create(ctx.get(), u64p(&p), CHEETAH_PROJECTIVE, &t_));
    }
    BasePointTable(const Context& ctx, const AffinePoint& p) : ctx_(ctx) {
        ctx.check(cheetah_basepoint_table_create(ctx.get(), u64p(&p), CHEETAH_AFFINE, &t_));
    }
    ~BasePointTable() {
        if (t_) cheetah_basepoint_table_destroy(t_);
    }
    BasePointTable(const BasePointTable&) = delete;
    BasePointTable& operator=(const BasePointTable&) = delete;
    std::vector<ProjectivePoint> multiply_batch(const std::vector<Scalar>& by) const {
        std::vector<ProjectivePoint> out(by.size());
        ctx_.check(cheetah_basepoint_table_multiply_batch(ctx_.get(), t_, by.data()->bytes.data(), u64p(out.data()), nullptr, by.size(),
                                                          CHEETAH_PROJECTIVE));
        return out;
    }

private:
    const Context& ctx_;
    cheetah_basepoint_table* t_ = nullptr;
};

// ProjectivePoint::multiply(&self, by), src/curve/projective.rs:614-625 (constant time)
inline std::vector<ProjectivePoint> multiply_batch(const Context& ctx, const std::vector<ProjectivePoint>& p,
                                                   const std::vector<Scalar>& by) {
    if (p.size() != by.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<ProjectivePoint> out(p.size());
    ctx.check(cheetah_multiply_batch(ctx.get(), u64p(p.data()), CHEETAH_PROJECTIVE, by.data()->bytes.data(), u64p(out.data()),
                                     nullptr, p.size(), CHEETAH_PROJECTIVE));
    return out;
}
// ProjectivePoint::multiply_vartime(&self, by), src/curve/projective.rs:633-662
inline std::vector<ProjectivePoint> multiply_vartime_batch(const Context& ctx, const std::vector<ProjectivePoint>& p,
                                                           const std::vector<Scalar>& by) {
    if (p.size() != by.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<ProjectivePoint> out(p.size());
    ctx.check(cheetah_multiply_vartime_batch(ctx.get(), u64p(p.data()), CHEETAH_PROJECTIVE, by.data()->bytes.data(),
                                             u64p(out.data()), nullptr, p.size(), CHEETAH_PROJECTIVE));
    return out;
}
// AffinePoint::multiply(&self, by), src/curve/affine.rs:373-375 (affine in, affine out)
inline std::vector<AffinePoint> multiply_batch(const Context& ctx, const std::vector<AffinePoint>& p,
                                               const std::vector<Scalar>& by, std::vector<uint8_t>& infinity) {
    if (p.size() != by.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<AffinePoint> out(p.size());
    infinity.assign(p.size(), 0);
    ctx.check(cheetah_multiply_batch(ctx.get(), u64p(p.data()), CHEETAH_AFFINE, by.data()->bytes.data(), u64p(out.data()),
                                     infinity.data(), p.size(), CHEETAH_AFFINE));
    return out;
}

// JacobianPoint::multiply(&self, by), src/curve/jacobian.rs:599-611 (constant time, Jacobian in / out)
inline std::vector<JacobianPoint> multiply_batch(const Context& ctx, const std::vector<JacobianPoint>& p,
                                                 const std::vector<Scalar>& by) {
    if (p.size() != by.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<JacobianPoint> out(p.size());
    ctx.check(cheetah_multiply_batch(ctx.get(), u64p(p.data()), CHEETAH_JACOBIAN, by.data()->bytes.data(), u64p(out.data()),
                                     nullptr, p.size(), CHEETAH_JACOBIAN));
    return out;
}
// JacobianPoint::multiply_vartime(&self, by), src/curve/jacobian.rs:619-649
inline std::vector<JacobianPoint> multiply_vartime_batch(const Context& ctx, const std::vector<JacobianPoint>& p,
                                                         const std::vector<Scalar>& by) {
    if (p.size() != by.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<JacobianPoint> out(p.size());
    ctx.check(cheetah_multiply_vartime_batch(ctx.get(), u64p(p.data()), CHEETAH_JACOBIAN, by.data()->bytes.data(),
                                             u64p(out.data()), nullptr, p.size(), CHEETAH_JACOBIAN));
    return out;
}
// &BASEPOINT_LOOKUP * &scalar, src/constants.rs:102-135 (Jacobian results)
struct BasePointLookupT {
    std::vector<JacobianPoint> multiply_batch(const Context& ctx, const std::vector<Scalar>& by) const {
        std::vector<JacobianPoint> out(by.size());
        ctx.check(cheetah_mul_by_generator_batch(ctx.get(), by.data()->bytes.data(), u64p(out.data()), nullptr, by.size(),
                                                 CHEETAH_JACOBIAN));
        return out;
    }
};
static const BasePointLookupT BASEPOINT_LOOKUP{};

// ProjectivePoint::multiply_many(points, scalars), src/curve/projective.rs:817-832, one call per group of m:
// out[g] = sum_j scalars[g*m + j] * points[g*m + j]; m = 2 is multiply_double (:666-685).
inline std::vector<ProjectivePoint> multiply_many_batch(const Context& ctx, const std::vector<ProjectivePoint>& points,
                                                        const std::vector<Scalar>& scalars, int m, bool vartime = false) {
    if (m < 1 || points.size() != scalars.size() || points.size() % (size_t)m) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<ProjectivePoint> out(points.size() / (size_t)m);
    auto f = vartime ? cheetah_multiply_many_vartime_batch : cheetah_multiply_many_batch;
    ctx.check(f(ctx.get(), u64p(points.data()), CHEETAH_PROJECTIVE, scalars.data()->bytes.data(), m, u64p(out.data()), nullptr,
                out.size(), CHEETAH_PROJECTIVE));
    return out;
}
// ProjectivePoint::multiply_double_with_basepoint_vartime, src/curve/projective.rs:756-809:
// out[i] = by_self[i] * points[i] + by_basepoint[i] * G
inline std::vector<ProjectivePoint> multiply_double_with_basepoint_vartime_batch(const Context& ctx,
                                                                                 const std::vector<ProjectivePoint>& points,
                                                                                 const std::vector<Scalar>& by_self,
                                                                                 const std::vector<Scalar>& by_basepoint) {
    if (points.size() != by_self.size() || points.size() != by_basepoint.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<Scalar> sc(2 * points.size());
    for (size_t i = 0; i < points.size(); i++) {
        sc[2 * i] = by_self[i];
        sc[2 * i + 1] = by_basepoint[i];
    }
    std::vector<ProjectivePoint> out(points.size());
    ctx.check(cheetah_multiply_double_with_basepoint_vartime_batch(ctx.get(), u64p(points.data()), CHEETAH_PROJECTIVE,
                                                                   sc.data()->bytes.data(), u64p(out.data()), nullptr, out.size(),
                                                                   CHEETAH_PROJECTIVE));
    return out;
}
// ProjectivePoint::is_torsion_free, src/curve/projective.rs:885-893
inline std::vector<uint8_t> is_torsion_free_batch(const Context& ctx, const std::vector<ProjectivePoint>& p) {
    std::vector<uint8_t> flags(p.size());
    ctx.check(cheetah_is_torsion_free_batch(ctx.get(), u64p(p.data()), CHEETAH_PROJECTIVE, flags.data(), p.size()));
    return flags;
}
// ProjectivePoint::clear_cofactor, src/curve/projective.rs:872-881
inline std::vector<ProjectivePoint> clear_cofactor_batch(const Context& ctx, const std::vector<ProjectivePoint>& p) {
    std::vector<ProjectivePoint> out(p.size());
    ctx.check(cheetah_clear_cofactor_batch(ctx.get(), u64p(p.data()), CHEETAH_PROJECTIVE, u64p(out.data()), nullptr, p.size(),
                                           CHEETAH_PROJECTIVE));
    return out;
}

// n points shaped like ProjectivePoint::random (src/curve/projective.rs:228-248) / AffinePoint::random
// (src/curve/affine.rs:243-245), from a seed instead of a caller's rng: element i of stream `seed` is a function of
// (seed, start + i) only (include/cheetah_hip.h).
inline std::vector<ProjectivePoint> random_projective_batch(const Context& ctx, uint64_t seed, size_t n, uint64_t start = 0) {
    std::vector<ProjectivePoint> out(n);
    ctx.check(cheetah_random_points(ctx.get(), seed, start, u64p(out.data()), nullptr, n, CHEETAH_PROJECTIVE));
    return out;
}
inline std::vector<AffinePoint> random_affine_batch(const Context& ctx, uint64_t seed, size_t n, std::vector<uint8_t>& infinity,
                                                    uint64_t start = 0) {
    std::vector<AffinePoint> out(n);
    infinity.assign(n, 0);
    ctx.check(cheetah_random_points(ctx.get(), seed, start, u64p(out.data()), infinity.data(), n, CHEETAH_AFFINE));
    return out;
}

// ---- complete group law over batches ------------------------------------------------
// &ProjectivePoint + &ProjectivePoint = ProjectivePoint::add, src/curve/projective.rs:456-515
inline std::vector<ProjectivePoint> add_batch(const Context& ctx, const std::vector<ProjectivePoint>& lhs,
                                              const std::vector<ProjectivePoint>& rhs) {
    if (lhs.size() != rhs.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<ProjectivePoint> out(lhs.size());
    ctx.check(cheetah_add_batch(ctx.get(), u64p(lhs.data()), CHEETAH_PROJECTIVE, u64p(rhs.data()), CHEETAH_PROJECTIVE, u64p(out.data()),
                                lhs.size()));
    return out;
}
// ProjectivePoint::add_mixed, src/curve/projective.rs:518-569 ((0, 1) is the affine identity)
inline std::vector<ProjectivePoint> add_mixed_batch(const Context& ctx, const std::vector<ProjectivePoint>& lhs,
                                                    const std::vector<AffinePoint>& rhs) {
    if (lhs.size() != rhs.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<ProjectivePoint> out(lhs.size());
    ctx.check(cheetah_add_batch(ctx.get(), u64p(lhs.data()), CHEETAH_PROJECTIVE, u64p(rhs.data()), CHEETAH_AFFINE, u64p(out.data()),
                                lhs.size()));
    return out;
}
// ProjectivePoint::double, src/curve/projective.rs:349-393
inline std::vector<ProjectivePoint> double_batch(const Context& ctx, const std::vector<ProjectivePoint>& p) {
    std::vector<ProjectivePoint> out(p.size());
    ctx.check(cheetah_double_batch(ctx.get(), u64p(p.data()), CHEETAH_PROJECTIVE, u64p(out.data()), p.size()));
    return out;
}

// ProjectivePoint::double_multi(k), src/curve/projective.rs:322-331
inline std::vector<ProjectivePoint> double_multi_batch(const Context& ctx, const std::vector<ProjectivePoint>& p, uint32_t k) {
    std::vector<ProjectivePoint> out(p.size());
    ctx.check(cheetah_double_multi_batch(ctx.get(), u64p(p.data()), CHEETAH_PROJECTIVE, k, u64p(out.data()), p.size()));
    return out;
}
// -&ProjectivePoint = ProjectivePoint::neg, src/curve/projective.rs:447-453
inline std::vector<ProjectivePoint> neg_batch(const Context& ctx, const std::vector<ProjectivePoint>& p) {
    std::vector<ProjectivePoint> out(p.size());
    ctx.check(cheetah_neg_batch(ctx.get(), u64p(p.data()), CHEETAH_PROJECTIVE, u64p(out.data()), p.size()));
    return out;
}
// ProjectivePoint == ProjectivePoint (ConstantTimeEq), src/curve/projective.rs:118-135
inline std::vector<uint8_t> eq_batch(const Context& ctx, const std::vector<ProjectivePoint>& a, const std::vector<ProjectivePoint>& b) {
    if (a.size() != b.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<uint8_t> flags(a.size());
    ctx.check(cheetah_eq_batch(ctx.get(), u64p(a.data()), u64p(b.data()), CHEETAH_PROJECTIVE, flags.data(), a.size()));
    return flags;
}

// ---- point encodings (src/curve/encoding.rs) ---------------------------------------
struct CompressedPoint {
    std::array<uint8_t, 49> bytes;  // src/curve/encoding.rs:29
};
struct UncompressedPoint {
    std::array<uint8_t, 97> bytes;  // src/curve/encoding.rs:153
};
// AffinePoint::to_compressed, src/curve/encoding.rs:55-77
inline std::vector<CompressedPoint> to_compressed_batch(const Context& ctx, const std::vector<AffinePoint>& p,
                                                        const std::vector<uint8_t>& infinity) {
    if (p.size() != infinity.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<CompressedPoint> out(p.size());
    ctx.check(cheetah_encode_compressed_batch(ctx.get(), u64p(p.data()), infinity.data(), out.data()->bytes.data(), p.size()));
    return out;
}
// AffinePoint::to_uncompressed, src/curve/encoding.rs:169-183
inline std::vector<UncompressedPoint> to_uncompressed_batch(const Context& ctx, const std::vector<AffinePoint>& p,
                                                            const std::vector<uint8_t>& infinity) {
    if (p.size() != infinity.size()) throw Error(CHEETAH_ERR_INVALID, "length mismatch");
    std::vector<UncompressedPoint> out(p.size());
    ctx.check(cheetah_encode_uncompressed_batch(ctx.get(), u64p(p.data()), infinity.data(), out.data()->bytes.data(), p.size()));
    return out;
}
// AffinePoint::from_compressed (checked, src/curve/affine.rs:303-309) / from_compressed_unchecked (:315-317);
// is_some[i] = 0 marks a CtOption::none
inline std::vector<AffinePoint> from_compressed_batch(const Context& ctx, const std::vector<CompressedPoint>& b,
                                                      std::vector<uint8_t>& infinity, std::vector<uint8_t>& is_some,
                                                      bool checked = true) {
    std::vector<AffinePoint> out(b.size());
    infinity.assign(b.size(), 0);
    is_some.assign(b.size(), 0);
    ctx.check(cheetah_decode_compressed_batch(ctx.get(), b.data()->bytes.data(), u64p(out.data()), infinity.data(), is_some.data(),
                                              b.size(), checked ? 1 : 0));
    return out;
}
// AffinePoint::from_uncompressed (checked, src/curve/affine.rs:289-292) / from_uncompressed_unchecked (:298-300)
inline std::vector<AffinePoint> from_uncompressed_batch(const Context& ctx, const std::vector<UncompressedPoint>& b,
                                                        std::vector<uint8_t>& infinity, std::vector<uint8_t>& is_some,
                                                        bool checked = true) {
    std::vector<AffinePoint> out(b.size());
    infinity.assign(b.size(), 0);
    is_some.assign(b.size(), 0);
    ctx.check(cheetah_decode_uncompressed_batch(ctx.get(), b.data()->bytes.data(), u64p(out.data()), infinity.data(),
                                                is_some.data(), b.size(), checked ? 1 : 0));
    return out;
}
// AffinePoint::from_x, src/curve/affine.rs:326-343
inline std::vector<AffinePoint> from_x_batch(const Context& ctx, const std::vector<Fp6>& x, std::vector<uint8_t>& infinity,
                                             std::vector<uint8_t>& is_some) {
    std::vector<AffinePoint> out(x.size());
    infinity.assign(x.size(), 0);
    is_some.assign(x.size(), 0);
    ctx.check(cheetah_affine_from_x_batch(ctx.get(), u64p(x.data()), u64p(out.data()), infinity.data(), is_some.data(), x.size()));
    return out;
}

}  // namespace cheetah
