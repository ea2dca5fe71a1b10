"""ctypes loader for the CPU oracle (oracle/libcheetah_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by cheetah_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcheetah_oracle.so")

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "cheetah_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(os.path.join(_HERE, "cheetah_oracle.h"))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcheetah_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    sig = {
        "ora_fp_add": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "ora_fp_sub": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "ora_fp_neg": (C.c_uint64, [C.c_uint64]),
        "ora_fp_double": (C.c_uint64, [C.c_uint64]),
        "ora_fp_triple": (C.c_uint64, [C.c_uint64]),
        "ora_fp_mul": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "ora_fp_square": (C.c_uint64, [C.c_uint64]),
        "ora_fp_mul_by_u32": (C.c_uint64, [C.c_uint64, C.c_uint32]),
        "ora_fp_canonical": (C.c_uint64, [C.c_uint64]),
        "ora_fp_invert": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint64)]),
        "ora_fp_exp": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "ora_fp_sqrt": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint64)]),
        "ora_fp6_exp": (None, [u64p, u64p, u64p]),
        "ora_fp6_exp_batch": (None, [u64p, u64p, u64p, C.c_size_t]),
        "ora_reduce_u128": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "ora_reduce_u96": (C.c_uint64, [C.c_uint64, C.c_uint32]),
        "ora_fp6_add": (None, [u64p, u64p, u64p]),
        "ora_fp6_sub": (None, [u64p, u64p, u64p]),
        "ora_fp6_neg": (None, [u64p, u64p]),
        "ora_fp6_double": (None, [u64p, u64p]),
        "ora_fp6_triple": (None, [u64p, u64p]),
        "ora_fp6_mul": (None, [u64p, u64p, u64p]),
        "ora_fp6_square": (None, [u64p, u64p]),
        "ora_fp6_mul_by_u32": (None, [u64p, C.c_uint32, u64p]),
        "ora_fp6_mul_by_fp": (None, [u64p, C.c_uint64, u64p]),
        "ora_fp6_frobenius": (None, [u64p, u64p]),
        "ora_fp6_frobenius_double": (None, [u64p, u64p]),
        "ora_fp6_frobenius_triple": (None, [u64p, u64p]),
        "ora_fp6_invert": (C.c_int, [u64p, u64p]),
        "ora_fp6_canonical": (None, [u64p, u64p]),
        "ora_fp6_is_zero": (C.c_int, [u64p]),
        "ora_mul_by_3b": (None, [u64p, u64p]),
        "ora_bytes_to_radix_16": (None, [u8p, i8p]),
        "ora_bytes_to_radix_256": (None, [u8p, i8p]),
        "ora_bytes_to_wnaf_vartime": (None, [u8p, C.c_int, i8p]),
        "ora_proj_double": (None, [u64p, u64p]),
        "ora_proj_double_unchecked": (None, [u64p, u64p]),
        "ora_proj_add": (None, [u64p, u64p, u64p]),
        "ora_proj_add_mixed": (None, [u64p, u64p, C.c_uint8, u64p]),
        "ora_proj_add_mixed_unchecked": (None, [u64p, u64p, C.c_uint8, u64p]),
        "ora_proj_to_affine": (None, [u64p, u64p, u8p]),
        "ora_proj_is_on_curve": (C.c_int, [u64p]),
        "ora_jac_double": (None, [u64p, u64p]),
        "ora_jac_double_unchecked": (None, [u64p, u64p]),
        "ora_jac_add_mixed_unchecked": (None, [u64p, u64p, C.c_uint8, u64p]),
        "ora_jac_add": (None, [u64p, u64p, u64p]),
        "ora_jac_add_mixed": (None, [u64p, u64p, C.c_uint8, u64p]),
        "ora_jac_eq": (C.c_int, [u64p, u64p]),
        "ora_proj_eq": (C.c_int, [u64p, u64p]),
        "ora_add_batch": (None, [u64p, C.c_int, u64p, C.c_int, u64p, C.c_size_t]),
        "ora_double_batch": (None, [u64p, C.c_int, u64p, C.c_size_t]),
        "ora_double_multi_batch": (None, [u64p, C.c_int, C.c_uint32, u64p, C.c_size_t]),
        "ora_neg_batch": (None, [u64p, C.c_int, u64p, C.c_size_t]),
        "ora_double_multi_unchecked_batch": (None, [u64p, C.c_int, C.c_uint32, u64p, C.c_size_t]),
        "ora_add_mixed_unchecked_batch": (None, [u64p, C.c_int, u64p, u64p, C.c_size_t]),
        "ora_eq_batch": (None, [u64p, u64p, C.c_int, u8p, C.c_size_t]),
        "ora_jac_to_affine": (None, [u64p, u64p, u8p]),
        "ora_jac_is_on_curve": (C.c_int, [u64p]),
        "ora_mjac_double_unchecked": (None, [u64p, u64p]),
        "ora_mjac_add_mixed_unchecked": (None, [u64p, u64p, C.c_uint8, u64p]),
        "ora_mjac_to_affine": (None, [u64p, u64p, u8p]),
        "ora_proj_batch_normalize": (None, [u64p, C.c_size_t, u64p, u8p]),
        "ora_jac_batch_normalize": (None, [u64p, C.c_size_t, u64p, u8p]),
        "ora_lookup_table_from_proj": (None, [u64p, C.c_int, u64p, u8p]),
        "ora_naf_table_from_proj": (None, [u64p, C.c_int, u64p, u8p]),
        "ora_basepoint_table": (C.POINTER(C.c_uint64), []),
        "ora_basepoint_lookup": (C.POINTER(C.c_uint64), []),
        "ora_minus_shift_point_array": (C.POINTER(C.c_uint64), []),
        "ora_shift_point_affine": (C.POINTER(C.c_uint64), []),
        "ora_generator_affine": (C.POINTER(C.c_uint64), []),
        "ora_proj_multiply": (None, [u64p, u8p, u64p]),
        "ora_proj_multiply_vartime": (None, [u64p, u8p, u64p]),
        "ora_basepoint_table_multiply": (None, [u8p, u64p]),
        "ora_basepoint_lookup_multiply": (None, [u8p, u64p]),
        "ora_jac_multiply": (None, [u64p, u8p, u64p]),
        "ora_jac_multiply_vartime": (None, [u64p, u8p, u64p]),
        "ora_num_threads": (C.c_int, []),
        "ora_basepoint_table_create": (None, [u64p, C.c_int, u64p, u8p]),
        "ora_basepoint_table_multiply_batch": (None, [u64p, u8p, u8p, u64p, u8p, C.c_size_t, C.c_int]),
        "ora_affine_is_on_curve": (C.c_int, [u64p]),
        "ora_fp6_mul_batch": (None, [u64p, u64p, u64p, C.c_size_t]),
        "ora_fp6_square_batch": (None, [u64p, u64p, C.c_size_t]),
        "ora_fp6_invert_batch": (None, [u64p, u64p, u8p, C.c_size_t]),
        "ora_mul_by_generator_batch": (None, [u8p, u64p, u8p, C.c_size_t, C.c_int]),
        "ora_multiply_batch": (None, [u64p, C.c_int, u8p, u64p, u8p, C.c_size_t, C.c_int]),
        "ora_multiply_vartime_batch": (None, [u64p, C.c_int, u8p, u64p, u8p, C.c_size_t, C.c_int]),
        "ora_odd_multiples_basepoint": (C.POINTER(C.c_uint64), []),
        "ora_fp6_lexicographically_largest": (C.c_int, [u64p]),
        "ora_fp3_mul": (C.c_int, [u64p, u64p, u64p]),
        "ora_fp3_sqrt": (C.c_int, [u64p, u64p]),
        "ora_fp6_sqrt": (C.c_int, [u64p, u64p]),
        "ora_fp6_sqrt_batch": (None, [u64p, u64p, u8p, C.c_size_t]),
        "ora_compressed_to_affine": (C.c_int, [u8p, C.c_int, u64p, u8p]),
        "ora_decode_compressed_batch": (None, [u8p, C.c_int, u64p, u8p, u8p, C.c_size_t]),
        "ora_affine_from_x": (C.c_int, [u64p, u64p, u8p]),
        "ora_affine_from_x_batch": (None, [u64p, u64p, u8p, u8p, C.c_size_t]),
        "ora_affine_to_compressed": (None, [u64p, C.c_uint8, u8p]),
        "ora_affine_to_uncompressed": (None, [u64p, C.c_uint8, u8p]),
        "ora_uncompressed_to_affine": (C.c_int, [u8p, C.c_int, u64p, u8p]),
        "ora_encode_batch": (None, [u64p, u8p, u8p, C.c_size_t, C.c_int]),
        "ora_decode_uncompressed_batch": (None, [u8p, C.c_int, u64p, u8p, u8p, C.c_size_t]),
        "ora_multiply_many_batch": (None, [u64p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, u64p, u8p, C.c_size_t, C.c_int]),
        "ora_jac_batch_normalize_chunked": (None, [u64p, C.c_size_t, C.c_size_t, u64p, u8p]),
        "ora_proj_batch_normalize_chunked": (None, [u64p, C.c_size_t, C.c_size_t, u64p, u8p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


# ---- numpy-friendly helpers -------------------------------------------------

def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _u8(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


def fp6_binop(name, a, b):
    out = np.zeros(6, np.uint64)
    getattr(lib(), name)(_u64(a), _u64(b), out)
    return out


def fp6_unop(name, a):
    out = np.zeros(6, np.uint64)
    getattr(lib(), name)(_u64(a), out)
    return out


def fp6_invert(a):
    out = np.zeros(6, np.uint64)
    ok = lib().ora_fp6_invert(_u64(a), out)
    return (out if ok else None)


def table(name, n_points):
    ptr = getattr(lib(), name)()
    return np.ctypeslib.as_array(ptr, shape=(n_points, 12)).copy()


def canon(a):
    return _u64(a) % np.uint64(0xFFFFFFFF00000001)


def fp6_mul_batch(a, b):
    a, b = _u64(a), _u64(b)
    out = np.empty_like(a)
    lib().ora_fp6_mul_batch(a, b, out, a.size // 6)
    return out


def fp6_square_batch(a):
    a = _u64(a)
    out = np.empty_like(a)
    lib().ora_fp6_square_batch(a, out, a.size // 6)
    return out


def fp6_invert_batch(a):
    a = _u64(a)
    out = np.empty_like(a)
    ok = np.empty(a.size // 6, np.uint8)
    lib().ora_fp6_invert_batch(a, out, ok, a.size // 6)
    return out, ok


def _out_bufs(n, out_form):
    w = 12 if out_form == 0 else 18
    return np.empty((n, w), np.uint64), np.empty(n, np.uint8)


def mul_by_generator_batch(scalars, out_form=0):
    s = _u8(scalars).reshape(-1, 32)
    out, inf = _out_bufs(len(s), out_form)
    lib().ora_mul_by_generator_batch(s, out, inf, len(s), out_form)
    return out, inf


def multiply_batch(pts, in_form, scalars, out_form=0, vartime=False):
    s = _u8(scalars).reshape(-1, 32)
    p = _u64(pts).reshape(len(s), 12 if in_form == 0 else 18)
    if {in_form, out_form} >= {1, 2}:
        raise ValueError("PROJECTIVE and JACOBIAN cannot be mixed")
    out, inf = _out_bufs(len(s), out_form)
    f = lib().ora_multiply_vartime_batch if vartime else lib().ora_multiply_batch
    f(p, in_form, s, out, inf, len(s), out_form)
    return out, inf


def multiply_many_batch(pts, in_form, scalars, m, out_form=0, vartime=False, with_basepoint=False):
    """out[g] = sum_j scalars[g, j] * pts[g, j] (+ scalars[g, m] * G when with_basepoint)."""
    ns = m + (1 if with_basepoint else 0)
    s = _u8(scalars).reshape(-1, ns * 32)
    n = len(s)
    p = _u64(pts).reshape(n * m, 12 if in_form == 0 else 18)
    out, inf = _out_bufs(n, out_form)
    lib().ora_multiply_many_batch(p, in_form, s, m, int(vartime), int(with_basepoint), out, inf, n, out_form)
    return out, inf


def encode_batch(xy, inf, compressed):
    xy, inf = _u64(xy).reshape(-1, 12), _u8(inf)
    out = np.empty((len(xy), 49 if compressed else 97), np.uint8)
    lib().ora_encode_batch(xy, inf, out, len(xy), int(compressed))
    return out


def decode_uncompressed_batch(b, checked):
    b = _u8(b).reshape(-1, 97)
    n = len(b)
    xy, inf, some = np.empty((n, 12), np.uint64), np.empty(n, np.uint8), np.empty(n, np.uint8)
    lib().ora_decode_uncompressed_batch(b, int(checked), xy, inf, some, n)
    return xy, inf, some


def decode_compressed_batch(b, checked):
    b = _u8(b).reshape(-1, 49)
    n = len(b)
    xy, inf, some = np.empty((n, 12), np.uint64), np.empty(n, np.uint8), np.empty(n, np.uint8)
    lib().ora_decode_compressed_batch(b, int(checked), xy, inf, some, n)
    return xy, inf, some


def affine_from_x_batch(x):
    x = _u64(x).reshape(-1, 6)
    n = len(x)
    xy, inf, some = np.empty((n, 12), np.uint64), np.empty(n, np.uint8), np.empty(n, np.uint8)
    lib().ora_affine_from_x_batch(x, xy, inf, some, n)
    return xy, inf, some


def fp6_sqrt_batch(a):
    a = _u64(a).reshape(-1, 6)
    out = np.empty_like(a)
    ok = np.empty(len(a), np.uint8)
    lib().ora_fp6_sqrt_batch(a, out, ok, len(a))
    return out, ok


def batch_normalize(p, jacobian, chunk=0):
    p = _u64(p).reshape(-1, 18)
    n = len(p)
    xy = np.empty((n, 12), np.uint64)
    inf = np.empty(n, np.uint8)
    f = lib().ora_jac_batch_normalize_chunked if jacobian else lib().ora_proj_batch_normalize_chunked
    f(p, n, chunk, xy, inf)
    return xy, inf


def basepoint_table_create(point, in_form=0):
    """BasePointTable::create(&P): (256 x 12 limbs, 256 identity flags), src/lookup.rs:164-173."""
    p = _u64(point).reshape(-1)
    xy = np.empty((256, 12), np.uint64)
    inf = np.empty(256, np.uint8)
    lib().ora_basepoint_table_create(p, in_form, xy, inf)
    return xy, inf


def basepoint_table_multiply_batch(table, scalars, out_form=1):
    """BasePointTable::multiply with a caller's table, src/lookup.rs:189-206."""
    txy, tinf = table
    s = _u8(scalars)
    n = len(s)
    out, inf = _out_bufs(n, out_form)
    lib().ora_basepoint_table_multiply_batch(_u64(txy), _u8(tinf), s,
                                             out, inf, n, out_form)
    return out, inf


def is_on_curve_batch(pts, in_form):
    p = _u64(pts)
    f = {0: lib().ora_affine_is_on_curve, 1: lib().ora_proj_is_on_curve, 2: lib().ora_jac_is_on_curve}[in_form]
    return np.array([f(p[i]) for i in range(len(p))], np.uint8)


def add_batch(lhs, lhs_form, rhs, rhs_form):
    """Complete additions: ProjectivePoint::add / add_mixed (src/curve/projective.rs:456-569),
    JacobianPoint::add / add_mixed (src/curve/jacobian.rs:452-552); the result in lhs_form."""
    a = _u64(lhs).reshape(-1, 18)
    b = _u64(rhs).reshape(len(a), 12 if rhs_form == 0 else 18)
    out = np.empty((len(a), 18), np.uint64)
    lib().ora_add_batch(a, lhs_form, b, rhs_form, out, len(a))
    return out


def double_batch(pts, form):
    """ProjectivePoint::double (src/curve/projective.rs:349-393) / JacobianPoint::double (src/curve/jacobian.rs:368-402)."""
    a = _u64(pts).reshape(-1, 18)
    out = np.empty_like(a)
    lib().ora_double_batch(a, form, out, len(a))
    return out


def double_multi_batch(pts, form, k):
    """::double_multi(k) (src/curve/projective.rs:322-331, src/curve/jacobian.rs:341-350)."""
    a = _u64(pts).reshape(-1, 18)
    out = np.empty_like(a)
    lib().ora_double_multi_batch(a, form, k, out, len(a))
    return out


def neg_batch(pts, form):
    a = _u64(pts).reshape(-1, 12 if form == 0 else 18)
    out = np.empty_like(a)
    lib().ora_neg_batch(a, form, out, len(a))
    return out


def eq_batch(a, b, form):
    a = _u64(a).reshape(-1, 12 if form == 0 else 18)
    b = _u64(b).reshape(a.shape)
    flags = np.empty(len(a), np.uint8)
    lib().ora_eq_batch(a, b, form, flags, len(a))
    return flags


def double_multi_unchecked_batch(pts, form, k):
    a = _u64(pts).reshape(-1, 18)
    out = np.empty_like(a)
    lib().ora_double_multi_unchecked_batch(a, form, k, out, len(a))
    return out


def add_mixed_unchecked_batch(lhs, form, rhs_xy):
    a = _u64(lhs).reshape(-1, 18)
    b = _u64(rhs_xy).reshape(len(a), 12)
    out = np.empty_like(a)
    lib().ora_add_mixed_unchecked_batch(a, form, b, out, len(a))
    return out


def fp6_exp_batch(a, power):
    """Fp6::exp over a batch, one exponent < 2^384 (src/fp6.rs:1053-1086)."""
    a = _u64(a).reshape(-1, 6)
    pw = np.array([(int(power) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)], np.uint64)
    out = np.empty_like(a)
    lib().ora_fp6_exp_batch(a, pw, out, len(a))
    return out


def fp_sqrt(a):
    """Fp::sqrt (src/fp.rs:199-229): (root, is_some) per element."""
    a = _u64(a).reshape(-1)
    out, some = np.zeros(len(a), np.uint64), np.zeros(len(a), np.uint8)
    r = C.c_uint64()
    for i, v in enumerate(a):
        some[i] = lib().ora_fp_sqrt(int(v), C.byref(r))
        out[i] = r.value
    return out, some
