"""ctypes binding of the C ABI (include/cheetah_hip.h).

The library is the product: if it cannot be loaded, or a call fails, this module
raises -- there is no CPU fallback anywhere in cheetah_amd.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHEETAH_HIP_LIB: another build of the SAME library (A/B timing of kernel variants under tools/); never a fallback
LIB_PATH = os.environ.get("CHEETAH_HIP_LIB") or os.path.join(_HERE, "lib", "libcheetah_hip.so")

AFFINE, PROJECTIVE, JACOBIAN = 0, 1, 2

OK = 0
ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_SCALAR, ERR_NOMEM = -1, -2, -3, -4, -5

# every symbol include/cheetah_hip.h declares: (restype, argtypes)
_vp, _sz, _i = C.c_void_p, C.c_size_t, C.c_int
SIGNATURES = {
    "cheetah_ctx_create": (C.c_int32, [C.POINTER(C.c_int), _i, C.POINTER(_vp)]),
    "cheetah_ctx_num_devices": (C.c_int32, [_vp]),
    "cheetah_debug_ctx_create_replicated": (C.c_int32, [_i, _i, C.POINTER(_vp)]),
    "cheetah_basepoint_table_create": (C.c_int32, [_vp, _vp, _i, C.POINTER(_vp)]),
    "cheetah_basepoint_table_destroy": (C.c_int32, [_vp]),
    "cheetah_basepoint_table_multiply_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "cheetah_basepoint_table_multiply_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "cheetah_basepoint_table_read": (C.c_int32, [_vp, _vp, _vp, _sz]),
    "cheetah_is_on_curve_batch": (C.c_int32, [_vp, _vp, _i, _vp, _sz]),
    "cheetah_debug_set_reference_schedule": (C.c_int32, [_vp, _i]),
    "cheetah_ctx_set_point_independent_tables": (C.c_int32, [_vp, _i]),
    "cheetah_add_batch": (C.c_int32, [_vp, _vp, _i, _vp, _i, _vp, _sz]),
    "cheetah_add_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _i, _vp, _sz]),
    "cheetah_double_batch": (C.c_int32, [_vp, _vp, _i, _vp, _sz]),
    "cheetah_double_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _sz]),
    "cheetah_double_multi_batch": (C.c_int32, [_vp, _vp, _i, C.c_uint32, _vp, _sz]),
    "cheetah_double_multi_batch_dev": (C.c_int32, [_vp, _vp, _i, C.c_uint32, _vp, _sz]),
    "cheetah_double_multi_unchecked_batch": (C.c_int32, [_vp, _vp, _i, C.c_uint32, _vp, _sz]),
    "cheetah_double_multi_unchecked_batch_dev": (C.c_int32, [_vp, _vp, _i, C.c_uint32, _vp, _sz]),
    "cheetah_add_mixed_unchecked_batch": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _sz]),
    "cheetah_add_mixed_unchecked_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _sz]),
    "cheetah_fp_op_batch": (C.c_int32, [_vp, _i, _vp, _vp, C.c_uint64, _vp, _vp, _sz]),
    "cheetah_fp_op_batch_dev": (C.c_int32, [_vp, _i, _vp, _vp, C.c_uint64, _vp, _vp, _sz]),
    "cheetah_fp6_op_batch": (C.c_int32, [_vp, _i, _vp, _vp, C.c_uint64, _vp, _sz]),
    "cheetah_fp6_op_batch_dev": (C.c_int32, [_vp, _i, _vp, _vp, C.c_uint64, _vp, _sz]),
    "cheetah_neg_batch": (C.c_int32, [_vp, _vp, _i, _vp, _sz]),
    "cheetah_neg_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _sz]),
    "cheetah_eq_batch": (C.c_int32, [_vp, _vp, _vp, _i, _vp, _sz]),
    "cheetah_eq_batch_dev": (C.c_int32, [_vp, _vp, _vp, _i, _vp, _sz]),
    "cheetah_is_on_curve_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _sz]),
    "cheetah_ctx_destroy": (C.c_int32, [_vp]),
    "cheetah_ctx_set_stream": (C.c_int32, [_vp, _vp]),
    "cheetah_ctx_use_own_stream": (C.c_int32, [_vp]),
    "cheetah_ctx_synchronize": (C.c_int32, [_vp]),
    "cheetah_ctx_set_reserved_cus": (C.c_int32, [_vp, _i]),
    "cheetah_ctx_set_launch_rounds": (C.c_int32, [_vp, _i]),
    "cheetah_last_error": (C.c_char_p, [_vp]),
    "cheetah_version": (C.c_char_p, []),
    "cheetah_fp6_exp_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_fp6_exp_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_random_points_dev": (C.c_int32, [_vp, C.c_uint64, C.c_uint64, _vp, _vp, _sz, _i]),
    "cheetah_random_points": (C.c_int32, [_vp, C.c_uint64, C.c_uint64, _vp, _vp, _sz, _i]),
    "cheetah_fp6_mul_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_fp6_square_batch": (C.c_int32, [_vp, _vp, _vp, _sz]),
    "cheetah_fp6_invert_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_fp6_sqrt_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_fp6_sqrt_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_fp6_mul_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_fp6_square_batch_dev": (C.c_int32, [_vp, _vp, _vp, _sz]),
    "cheetah_fp6_invert_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_batch_normalize_projective": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_batch_normalize_jacobian": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_batch_normalize_projective_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_batch_normalize_jacobian_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_mul_by_generator_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _sz, _i]),
    "cheetah_mul_by_generator_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz, _i]),
    "cheetah_multiply_batch": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _i]),
    "cheetah_multiply_vartime_batch": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _i]),
    "cheetah_multiply_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _i]),
    "cheetah_multiply_vartime_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _i]),
    "cheetah_multiply_batch_dev_noalloc": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _i]),
    "cheetah_multiply_many_batch": (C.c_int32, [_vp, _vp, _i, _vp, _i, _vp, _vp, _sz, _i]),
    "cheetah_multiply_many_vartime_batch": (C.c_int32, [_vp, _vp, _i, _vp, _i, _vp, _vp, _sz, _i]),
    "cheetah_multiply_double_with_basepoint_vartime_batch": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _i]),
    "cheetah_multiply_many_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _i, _vp, _vp, _sz, _i]),
    "cheetah_multiply_many_vartime_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _i, _vp, _vp, _sz, _i]),
    "cheetah_multiply_double_with_basepoint_vartime_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _i]),
    "cheetah_is_torsion_free_batch": (C.c_int32, [_vp, _vp, _i, _vp, _sz]),
    "cheetah_clear_cofactor_batch": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _sz, _i]),
    "cheetah_is_torsion_free_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _sz]),
    "cheetah_clear_cofactor_batch_dev": (C.c_int32, [_vp, _vp, _i, _vp, _vp, _sz, _i]),
    "cheetah_encode_uncompressed_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_encode_compressed_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_decode_uncompressed_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "cheetah_decode_compressed_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "cheetah_decode_compressed_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "cheetah_affine_from_x_batch": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "cheetah_affine_from_x_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "cheetah_encode_uncompressed_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_encode_compressed_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _sz]),
    "cheetah_decode_uncompressed_batch_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "cheetah_read_table": (C.c_int32, [_vp, _i, _vp, _sz]),
    "cheetah_synth_fp6_dev": (C.c_int32, [_vp, C.c_uint64, C.c_uint64, _vp, _sz]),
    "cheetah_synth_scalars_dev": (C.c_int32, [_vp, C.c_uint64, C.c_uint64, _vp, _sz]),
    "cheetah_debug_op_dev": (C.c_int32, [_vp, _i, _vp, _vp, _vp, _sz]),
    "cheetah_calibrate_mad_peak": (C.c_int32, [_vp, C.POINTER(C.c_double)]),
    "cheetah_multiply_units_per_launch": (C.c_int32, [_vp, _sz, _i, _i, C.POINTER(C.c_size_t)]),
    "cheetah_timer_start": (C.c_int32, [_vp]),
    "cheetah_timer_stop_ms": (C.c_int32, [_vp, C.POINTER(C.c_float)]),
}

_lib = None


class CheetahError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cheetah_hip error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libcheetah_hip.so (built by `make -C cheetah_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own HIP/HSA runtime under the same SONAME.  Whichever copy is
        # loaded first serves the whole process, and torch tensors / streams handed to this
        # library must come from the SAME runtime, so let torch's load first when it is there.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP library missing: %s (run __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib
