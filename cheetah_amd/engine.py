"""Host-side mirror of the reference's API for the batched hot path.

Names follow the reference crate (src/lib.rs:77-94): `Fp6`, `ProjectivePoint`,
`JacobianPoint`, `BasePointTable` / `BASEPOINT_TABLE`; each batch method is the
array form of the reference method of the same name and returns, element by
element, that method's result in canonical limbs.  Everything is computed by the
HIP library through the C ABI (include/cheetah_hip.h); nothing here computes
field or curve arithmetic on the CPU.

Two kinds of operands are accepted everywhere:
  * numpy arrays (host memory)      -> the host-pointer entry points (PCIe included)
  * torch CUDA tensors (device HBM) -> the `_dev` entry points, asynchronous on
                                       torch's current stream
Layouts are the ABI's: Fp6 (n, 6) uint64; affine (n, 12) uint64 + (n,) uint8
infinity flags; projective / Jacobian (n, 18) uint64; scalars (n, 32) uint8.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import AFFINE, JACOBIAN, PROJECTIVE, CheetahError  # noqa: F401

try:  # torch is plumbing for device memory and streams only
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


class Engine:
    """One context (cheetah_ctx) on one GPU, or on several: `device` is a device id or a list of ids.
    Builds BASEPOINT_TABLE and the shift points on every device at construction
    (src/constants.rs:21-22, :43-90).  With several devices every call shards its batch contiguously
    over them and gathers the results (host arrays, or tensors of the first device).
    `replicate=k` (tests only) puts k shards on ONE device to exercise that path on a one-GPU box."""

    def __init__(self, device=0, replicate=0):
        self._lib = _capi.load()
        ids = [int(d) for d in device] if isinstance(device, (list, tuple)) else [int(device)]
        self.devices = ids
        self.device = ids[0] if ids else 0
        h = C.c_void_p()
        if replicate:
            rc = self._lib.cheetah_debug_ctx_create_replicated(self.device, int(replicate), C.byref(h))
        else:
            rc = self._lib.cheetah_ctx_create((C.c_int * len(ids))(*ids), len(ids), C.byref(h))
        if rc != 0:
            raise CheetahError(rc, self._lib.cheetah_last_error(None).decode())
        self._h = h

    @property
    def num_devices(self):
        return self._lib.cheetah_ctx_num_devices(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cheetah_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -----------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise CheetahError(rc, self._lib.cheetah_last_error(self._h).decode())

    def _use_torch_stream(self):
        self._check(self._lib.cheetah_ctx_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def synchronize(self):
        self._check(self._lib.cheetah_ctx_synchronize(self._h))

    @staticmethod
    def _np(a, dtype, width):
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.ndim != 2 or a.shape[1] != width:
            raise ValueError("expected shape (n, %d), got %r" % (width, a.shape))
        return a

    def _dev(self, t, itemsize, width):
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError("tensor must live on cuda:%d" % self.device)
        if t.element_size() != itemsize or not t.is_contiguous() or t.dim() != 2 or t.shape[1] != width:
            raise ValueError("expected a contiguous (n, %d) tensor with %d-byte elements" % (width, itemsize))
        return t

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device="cuda:%d" % self.device)

    # -- Fp6 ------------------------------------------------------------------
    def fp6_mul_batch(self, a, b):
        """out[i] = a[i] * b[i]   (Fp6::mul, src/fp6.rs:651-834)"""
        if _is_torch(a):
            a, b = self._dev(a, 8, 6), self._dev(b, 8, 6)
            if a.shape != b.shape:
                raise ValueError("shape mismatch")
            out = self._empty(a.shape, torch.uint64)
            self._use_torch_stream()
            self._check(self._lib.cheetah_fp6_mul_batch_dev(self._h, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0]))
            return out
        a, b = self._np(a, np.uint64, 6), self._np(b, np.uint64, 6)
        if a.shape != b.shape:
            raise ValueError("shape mismatch")
        out = np.empty_like(a)
        self._check(self._lib.cheetah_fp6_mul_batch(self._h, a.ctypes.data, b.ctypes.data, out.ctypes.data, len(a)))
        return out

    def fp6_square_batch(self, a):
        """out[i] = a[i]^2   (Fp6::square, src/fp6.rs:838-920)"""
        if _is_torch(a):
            a = self._dev(a, 8, 6)
            out = self._empty(a.shape, torch.uint64)
            self._use_torch_stream()
            self._check(self._lib.cheetah_fp6_square_batch_dev(self._h, a.data_ptr(), out.data_ptr(), a.shape[0]))
            return out
        a = self._np(a, np.uint64, 6)
        out = np.empty_like(a)
        self._check(self._lib.cheetah_fp6_square_batch(self._h, a.ctypes.data, out.ctypes.data, len(a)))
        return out

    FP_OPS = ("add", "sub", "neg", "double", "triple", "mul", "square", "mul_by_u32", "invert", "canonical", "sqrt", "exp")

    def fp_op_batch(self, op, a, b=None, scalar=0):
        """out[i] = op(a[i] [, b[i] | scalar]) in the base field, one u64 per element (src/fp.rs:125-196, :335-359); op is one
        of Engine.FP_OPS.  "invert" and "sqrt" (Fp::sqrt, src/fp.rs:199-229) return (out, is_some); "exp" (Fp::exp, :304-333) raises
        to the power `scalar`."""
        code = self.FP_OPS.index(op)
        binary = op in ("add", "sub", "mul")
        if binary and b is None:
            raise ValueError("%s needs two operands" % op)
        if _is_torch(a):
            if a.dim() != 1 or a.element_size() != 8 or not a.is_contiguous() or not a.is_cuda:
                raise ValueError("expected a contiguous 1-D CUDA tensor of 8-byte elements")
            bp = b.data_ptr() if binary else 0
            if binary and b.shape != a.shape:
                raise ValueError("shape mismatch")
            out = self._empty(a.shape, torch.uint64)
            some = self._empty(a.shape, torch.uint8) if op in ("invert", "sqrt") else None
            self._use_torch_stream()
            self._check(self._lib.cheetah_fp_op_batch_dev(self._h, code, a.data_ptr(), bp, int(scalar), out.data_ptr(),
                                                          some.data_ptr() if some is not None else 0, a.shape[0]))
            return (out, some) if some is not None else out
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
        bp = 0
        if binary:
            b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1)
            if a.shape != b.shape:
                raise ValueError("shape mismatch")
            bp = b.ctypes.data
        out = np.empty_like(a)
        some = np.empty(len(a), np.uint8) if op in ("invert", "sqrt") else None
        self._check(self._lib.cheetah_fp_op_batch(self._h, code, a.ctypes.data, bp, int(scalar), out.ctypes.data,
                                                  some.ctypes.data if some is not None else 0, len(a)))
        return (out, some) if some is not None else out

    FP6_OPS = ("add", "sub", "neg", "double", "triple", "mul_by_u32", "mul_by_fp", "canonical", "frobenius", "frobenius_double",
               "frobenius_triple", "mul_by_3b")

    def fp6_op_batch(self, op, a, b=None, scalar=0):
        """out[i] = op(a[i] [, b[i] | scalar]) for the coefficient-wise and structured Fp6 operations (src/fp6.rs:373-382,
        :428-464, :968-1029, :1088-1123; mul_by_3b: src/curve/mod.rs:54-106); op is one of Engine.FP6_OPS."""
        code = self.FP6_OPS.index(op)
        binary = op in ("add", "sub")
        if binary and b is None:
            raise ValueError("%s needs two operands" % op)
        if _is_torch(a):
            a = self._dev(a, 8, 6)
            bp = 0
            if binary:
                b = self._dev(b, 8, 6)
                if a.shape != b.shape:
                    raise ValueError("shape mismatch")
                bp = b.data_ptr()
            out = self._empty(a.shape, torch.uint64)
            self._use_torch_stream()
            self._check(self._lib.cheetah_fp6_op_batch_dev(self._h, code, a.data_ptr(), bp, int(scalar), out.data_ptr(), a.shape[0]))
            return out
        a = self._np(a, np.uint64, 6)
        bp = 0
        if binary:
            b = self._np(b, np.uint64, 6)
            if a.shape != b.shape:
                raise ValueError("shape mismatch")
            bp = b.ctypes.data
        out = np.empty_like(a)
        self._check(self._lib.cheetah_fp6_op_batch(self._h, code, a.ctypes.data, bp, int(scalar), out.ctypes.data, len(a)))
        return out

    def fp6_exp_batch(self, a, power):
        """out[i] = a[i]^power for one exponent < 2^384 (Fp6::exp / exp_vartime, src/fp6.rs:1053-1086)."""
        pw = (C.c_uint64 * 6)(*[(int(power) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)])
        if _is_torch(a):
            x = self._dev(a, 8, 6)
            out = self._empty(x.shape, torch.uint64)
            self._use_torch_stream()
            self._check(self._lib.cheetah_fp6_exp_batch_dev(self._h, x.data_ptr(), C.addressof(pw), out.data_ptr(), x.shape[0]))
            return out
        x = self._np(a, np.uint64, 6)
        out = np.empty_like(x)
        self._check(self._lib.cheetah_fp6_exp_batch(self._h, x.ctypes.data, C.addressof(pw), out.ctypes.data, len(x)))
        return out

    def random_points(self, seed, n, start=0, out_form=AFFINE, host=False):
        """n seeded points of the prime-order subgroup shaped like {Affine,Projective,Jacobian}Point::random
        (src/curve/projective.rs:228-248): x drawn, y = +-sqrt(x^3 + x + B), cofactor cleared.  Device tensors (out, inf),
        or numpy arrays with host=True (the same stream through the host-pointer entry point)."""
        w = self._out_width(out_form)
        if host:
            out, inf = np.empty((n, w), np.uint64), np.empty(n, np.uint8)
            self._check(self._lib.cheetah_random_points(self._h, int(seed), int(start), out.ctypes.data, inf.ctypes.data, n, out_form))
            return out, inf
        out, inf = self._empty((n, w), torch.uint64), self._empty((n,), torch.uint8)
        self._use_torch_stream()
        self._check(self._lib.cheetah_random_points_dev(self._h, int(seed), int(start), out.data_ptr(), inf.data_ptr(), n, out_form))
        return out, inf

    def _fp6_option(self, name, a):
        if _is_torch(a):
            a = self._dev(a, 8, 6)
            out = self._empty(a.shape, torch.uint64)
            some = self._empty((a.shape[0],), torch.uint8)
            self._use_torch_stream()
            self._check(getattr(self._lib, name + "_dev")(self._h, a.data_ptr(), out.data_ptr(), some.data_ptr(), a.shape[0]))
            return out, some
        a = self._np(a, np.uint64, 6)
        out = np.empty_like(a)
        some = np.empty(len(a), np.uint8)
        self._check(getattr(self._lib, name)(self._h, a.ctypes.data, out.ctypes.data, some.ctypes.data, len(a)))
        return out, some

    def fp6_invert_batch(self, a):
        """(out, is_some): out[i] = a[i]^-1, is_some[i] = 0 and out[i] = 0 for zero
        (Fp6::invert -> CtOption, src/fp6.rs:1035-1049)"""
        return self._fp6_option("cheetah_fp6_invert_batch", a)

    def fp6_sqrt_batch(self, a):
        """(out, is_some): out[i] = the root Fp6::sqrt returns (src/fp6.rs:922-964), is_some[i] = 0 and
        out[i] = 0 for a non-square"""
        return self._fp6_option("cheetah_fp6_sqrt_batch", a)

    # -- batch_normalize --------------------------------------------------------
    def batch_normalize(self, xyz, form):
        """(xy, infinity) for projective (form = PROJECTIVE, src/curve/projective.rs:897-930)
        or Jacobian (form = JACOBIAN, src/curve/jacobian.rs:886-920) points."""
        if form not in (PROJECTIVE, JACOBIAN):
            raise ValueError("form must be PROJECTIVE or JACOBIAN")
        jac = form == JACOBIAN
        if _is_torch(xyz):
            xyz = self._dev(xyz, 8, 18)
            n = xyz.shape[0]
            xy = self._empty((n, 12), torch.uint64)
            inf = self._empty((n,), torch.uint8)
            self._use_torch_stream()
            f = self._lib.cheetah_batch_normalize_jacobian_dev if jac else self._lib.cheetah_batch_normalize_projective_dev
            self._check(f(self._h, xyz.data_ptr(), xy.data_ptr(), inf.data_ptr(), n))
            return xy, inf
        xyz = self._np(xyz, np.uint64, 18)
        n = len(xyz)
        xy = np.empty((n, 12), np.uint64)
        inf = np.empty(n, np.uint8)
        f = self._lib.cheetah_batch_normalize_jacobian if jac else self._lib.cheetah_batch_normalize_projective
        self._check(f(self._h, xyz.ctypes.data, xy.ctypes.data, inf.ctypes.data, n))
        return xy, inf

    # -- ladders ------------------------------------------------------------------
    @staticmethod
    def _out_width(out_form):
        if out_form == AFFINE:
            return 12
        if out_form in (PROJECTIVE, JACOBIAN):
            return 18
        raise ValueError("form must be AFFINE, PROJECTIVE or JACOBIAN")

    def mul_by_generator_batch(self, scalars, out_form=AFFINE, out=None, inf=None):
        """out[i] = scalars[i] * G.  AFFINE / PROJECTIVE: &BASEPOINT_TABLE * &Scalar =
        BasePointTable::multiply (src/lookup.rs:189-206, src/constants.rs:92-98);
        JACOBIAN: &BASEPOINT_LOOKUP * &Scalar (src/constants.rs:102-135)."""
        w = self._out_width(out_form)
        if _is_torch(scalars):
            s = self._dev(scalars, 1, 32)
            n = s.shape[0]
            out = self._empty((n, w), torch.uint64) if out is None else self._dev(out, 8, w)
            inf = self._empty((n,), torch.uint8) if inf is None else inf
            self._use_torch_stream()
            self._check(self._lib.cheetah_mul_by_generator_batch_dev(self._h, s.data_ptr(), out.data_ptr(), inf.data_ptr(), n, out_form))
            return out, inf
        s = self._np(scalars, np.uint8, 32)
        n = len(s)
        out = np.empty((n, w), np.uint64)
        inf = np.empty(n, np.uint8)
        self._check(self._lib.cheetah_mul_by_generator_batch(self._h, s.ctypes.data, out.ctypes.data, inf.ctypes.data, n, out_form))
        return out, inf

    def _multiply(self, vartime, pts, scalars, in_form, out_form, out, inf):
        w = self._out_width(out_form)
        wi = self._out_width(in_form)
        if _is_torch(pts):
            p, s = self._dev(pts, 8, wi), self._dev(scalars, 1, 32)
            n = s.shape[0]
            if p.shape[0] != n:
                raise ValueError("points / scalars length mismatch")
            out = self._empty((n, w), torch.uint64) if out is None else self._dev(out, 8, w)
            inf = self._empty((n,), torch.uint8) if inf is None else inf
            self._use_torch_stream()
            f = self._lib.cheetah_multiply_vartime_batch_dev if vartime else self._lib.cheetah_multiply_batch_dev
            self._check(f(self._h, p.data_ptr(), in_form, s.data_ptr(), out.data_ptr(), inf.data_ptr(), n, out_form))
            return out, inf
        p, s = self._np(pts, np.uint64, wi), self._np(scalars, np.uint8, 32)
        n = len(s)
        if len(p) != n:
            raise ValueError("points / scalars length mismatch")
        out = np.empty((n, w), np.uint64)
        inf = np.empty(n, np.uint8)
        f = self._lib.cheetah_multiply_vartime_batch if vartime else self._lib.cheetah_multiply_batch
        self._check(f(self._h, p.ctypes.data, in_form, s.ctypes.data, out.ctypes.data, inf.ctypes.data, n, out_form))
        return out, inf

    def multiply_batch(self, pts, scalars, in_form=AFFINE, out_form=AFFINE, out=None, inf=None):
        """out[i] = scalars[i] * pts[i], constant time.  PROJECTIVE on either side:
        ProjectivePoint::multiply (src/curve/projective.rs:614-625); otherwise JacobianPoint::multiply
        (src/curve/jacobian.rs:599-611), which is also what AffinePoint::multiply runs (src/curve/affine.rs:373-375)."""
        return self._multiply(False, pts, scalars, in_form, out_form, out, inf)

    def multiply_vartime_batch(self, pts, scalars, in_form=AFFINE, out_form=AFFINE, out=None, inf=None):
        """out[i] = scalars[i] * pts[i], variable time.  ProjectivePoint::multiply_vartime
        (src/curve/projective.rs:633-662) or JacobianPoint / AffinePoint::multiply_vartime
        (src/curve/jacobian.rs:619-649), chosen by the forms as in multiply_batch."""
        return self._multiply(True, pts, scalars, in_form, out_form, out, inf)

    def multiply_batch_noalloc(self, pts, scalars, in_form=AFFINE, out_form=AFFINE, out=None, inf=None):
        """multiply_batch on device tensors through the entry point that allocates nothing
        (cheetah_multiply_batch_dev_noalloc): `out` and `inf` must be given."""
        w, wi = self._out_width(out_form), self._out_width(in_form)
        if out is None or inf is None:
            raise ValueError("multiply_batch_noalloc allocates nothing: pass `out` and `inf` (device tensors)")
        p, s = self._dev(pts, 8, wi), self._dev(scalars, 1, 32)
        out = self._dev(out, 8, w)
        if not _is_torch(inf) or inf.dtype != torch.uint8 or not inf.is_contiguous() or inf.numel() != s.shape[0]:
            raise ValueError("inf must be a contiguous uint8 device tensor with one byte per element")
        if out.shape[0] != s.shape[0] or p.shape[0] != s.shape[0]:
            raise ValueError("pts, scalars and out must hold the same number of elements")
        self._use_torch_stream()
        self._check(self._lib.cheetah_multiply_batch_dev_noalloc(self._h, p.data_ptr(), in_form, s.data_ptr(), out.data_ptr(),
                                                                 inf.data_ptr(), s.shape[0], out_form))
        return out, inf

    def basepoint_table(self, point, in_form=AFFINE):
        """BasePointTable::create(&P) (src/lookup.rs:164-173) for one caller-supplied point (host array)."""
        return BasePointTable(self, point, in_form)

    def is_on_curve_batch(self, pts, in_form=AFFINE):
        """flags[i] = pts[i].is_on_curve() (src/curve/affine.rs:364-368, projective.rs:940-946, jacobian.rs:930-938)"""
        wi = self._out_width(in_form)
        if _is_torch(pts):
            p = self._dev(pts, 8, wi)
            flags = self._empty((p.shape[0],), torch.uint8)
            self._use_torch_stream()
            self._check(self._lib.cheetah_is_on_curve_batch_dev(self._h, p.data_ptr(), in_form, flags.data_ptr(), p.shape[0]))
            return flags
        p = self._np(pts, np.uint64, wi)
        flags = np.empty(len(p), np.uint8)
        self._check(self._lib.cheetah_is_on_curve_batch(self._h, p.ctypes.data, in_form, flags.ctypes.data, len(p)))
        return flags

    def add_batch(self, lhs, rhs, lhs_form=JACOBIAN, rhs_form=None, out=None):
        """out[i] = lhs[i] + rhs[i] by the complete formulas, in lhs_form (PROJECTIVE or JACOBIAN).  rhs_form == lhs_form
        (the default): ProjectivePoint::add (src/curve/projective.rs:456-515) / JacobianPoint::add
        (src/curve/jacobian.rs:452-505); rhs_form == AFFINE: ::add_mixed (projective.rs:518-569, jacobian.rs:508-552)."""
        rhs_form = lhs_form if rhs_form is None else rhs_form
        wr = self._out_width(rhs_form)
        if _is_torch(lhs):
            a, b = self._dev(lhs, 8, 18), self._dev(rhs, 8, wr)
            if a.shape[0] != b.shape[0]:
                raise ValueError("lhs and rhs must hold the same number of points")
            out = self._empty((a.shape[0], 18), torch.uint64) if out is None else self._dev(out, 8, 18)
            self._use_torch_stream()
            self._check(self._lib.cheetah_add_batch_dev(self._h, a.data_ptr(), lhs_form, b.data_ptr(), rhs_form, out.data_ptr(),
                                                        a.shape[0]))
            return out
        a, b = self._np(lhs, np.uint64, 18), self._np(rhs, np.uint64, wr)
        if len(a) != len(b):
            raise ValueError("lhs and rhs must hold the same number of points")
        out = np.empty((len(a), 18), np.uint64)
        self._check(self._lib.cheetah_add_batch(self._h, a.ctypes.data, lhs_form, b.ctypes.data, rhs_form, out.ctypes.data, len(a)))
        return out

    def double_batch(self, pts, form=JACOBIAN, out=None):
        """out[i] = 2 pts[i]: ProjectivePoint::double (src/curve/projective.rs:349-393) or JacobianPoint::double
        (src/curve/jacobian.rs:368-402)."""
        if _is_torch(pts):
            a = self._dev(pts, 8, 18)
            out = self._empty((a.shape[0], 18), torch.uint64) if out is None else self._dev(out, 8, 18)
            self._use_torch_stream()
            self._check(self._lib.cheetah_double_batch_dev(self._h, a.data_ptr(), form, out.data_ptr(), a.shape[0]))
            return out
        a = self._np(pts, np.uint64, 18)
        out = np.empty((len(a), 18), np.uint64)
        self._check(self._lib.cheetah_double_batch(self._h, a.ctypes.data, form, out.ctypes.data, len(a)))
        return out

    def double_multi_batch(self, pts, k, form=JACOBIAN, out=None):
        """out[i] = 2^k pts[i]: ::double_multi(k) (src/curve/projective.rs:322-331, src/curve/jacobian.rs:341-350)."""
        if _is_torch(pts):
            a = self._dev(pts, 8, 18)
            out = self._empty((a.shape[0], 18), torch.uint64) if out is None else self._dev(out, 8, 18)
            self._use_torch_stream()
            self._check(self._lib.cheetah_double_multi_batch_dev(self._h, a.data_ptr(), form, int(k), out.data_ptr(), a.shape[0]))
            return out
        a = self._np(pts, np.uint64, 18)
        out = np.empty((len(a), 18), np.uint64)
        self._check(self._lib.cheetah_double_multi_batch(self._h, a.ctypes.data, form, int(k), out.ctypes.data, len(a)))
        return out

    def double_multi_unchecked_batch(self, pts, k, form=JACOBIAN):
        """::double_multi_unchecked(k) (src/curve/projective.rs:337-346, src/curve/jacobian.rs:356-365); k = 1: ::double_unchecked."""
        if _is_torch(pts):
            a = self._dev(pts, 8, 18)
            out = self._empty((a.shape[0], 18), torch.uint64)
            self._use_torch_stream()
            self._check(self._lib.cheetah_double_multi_unchecked_batch_dev(self._h, a.data_ptr(), form, int(k), out.data_ptr(), a.shape[0]))
            return out
        a = self._np(pts, np.uint64, 18)
        out = np.empty((len(a), 18), np.uint64)
        self._check(self._lib.cheetah_double_multi_unchecked_batch(self._h, a.ctypes.data, form, int(k), out.ctypes.data, len(a)))
        return out

    def add_mixed_unchecked_batch(self, lhs, rhs_xy, form=JACOBIAN):
        """::add_mixed_unchecked (src/curve/projective.rs:576-610, src/curve/jacobian.rs:559-595)."""
        if _is_torch(lhs):
            a, b = self._dev(lhs, 8, 18), self._dev(rhs_xy, 8, 12)
            if a.shape[0] != b.shape[0]:
                raise ValueError("lhs and rhs must hold the same number of points")
            out = self._empty((a.shape[0], 18), torch.uint64)
            self._use_torch_stream()
            self._check(self._lib.cheetah_add_mixed_unchecked_batch_dev(self._h, a.data_ptr(), form, b.data_ptr(), out.data_ptr(), a.shape[0]))
            return out
        a, b = self._np(lhs, np.uint64, 18), self._np(rhs_xy, np.uint64, 12)
        if len(a) != len(b):
            raise ValueError("lhs and rhs must hold the same number of points")
        out = np.empty((len(a), 18), np.uint64)
        self._check(self._lib.cheetah_add_mixed_unchecked_batch(self._h, a.ctypes.data, form, b.ctypes.data, out.ctypes.data, len(a)))
        return out

    def neg_batch(self, pts, form=AFFINE):
        """out[i] = -pts[i] (src/curve/affine.rs:164-175, projective.rs:447-453, jacobian.rs:443-449)."""
        w = self._out_width(form)
        if _is_torch(pts):
            a = self._dev(pts, 8, w)
            out = self._empty((a.shape[0], w), torch.uint64)
            self._use_torch_stream()
            self._check(self._lib.cheetah_neg_batch_dev(self._h, a.data_ptr(), form, out.data_ptr(), a.shape[0]))
            return out
        a = self._np(pts, np.uint64, w)
        out = np.empty((len(a), w), np.uint64)
        self._check(self._lib.cheetah_neg_batch(self._h, a.ctypes.data, form, out.ctypes.data, len(a)))
        return out

    def eq_batch(self, a, b, form=AFFINE):
        """flags[i] = a[i] == b[i] as points: ct_eq (src/curve/affine.rs:132-144, projective.rs:118-135, jacobian.rs:134-154)."""
        w = self._out_width(form)
        if _is_torch(a):
            x, y = self._dev(a, 8, w), self._dev(b, 8, w)
            if x.shape != y.shape:
                raise ValueError("shape mismatch")
            flags = self._empty((x.shape[0],), torch.uint8)
            self._use_torch_stream()
            self._check(self._lib.cheetah_eq_batch_dev(self._h, x.data_ptr(), y.data_ptr(), form, flags.data_ptr(), x.shape[0]))
            return flags
        x, y = self._np(a, np.uint64, w), self._np(b, np.uint64, w)
        if x.shape != y.shape:
            raise ValueError("shape mismatch")
        flags = np.empty(len(x), np.uint8)
        self._check(self._lib.cheetah_eq_batch(self._h, x.ctypes.data, y.ctypes.data, form, flags.ctypes.data, len(x)))
        return flags

    # -- multi-scalar ladders, subgroup check, cofactor clearing ----------------------
    def _many(self, fname, pts, scalars, m, ns, in_form, out_form):
        w, wi = self._out_width(out_form), self._out_width(in_form)
        if _is_torch(pts):
            p, s = self._dev(pts, 8, wi), self._dev(scalars, 1, 32)
            if s.shape[0] % ns or p.shape[0] * ns != s.shape[0] * m:
                raise ValueError("need m points and %d scalars per group" % ns)
            n = s.shape[0] // ns
            out, inf = self._empty((n, w), torch.uint64), self._empty((n,), torch.uint8)
            self._use_torch_stream()
            f = getattr(self._lib, fname + "_dev")
            args = (self._h, p.data_ptr(), in_form, s.data_ptr()) + ((m,) if "many" in fname else ()) + (out.data_ptr(), inf.data_ptr(), n, out_form)
            self._check(f(*args))
            return out, inf
        p, s = self._np(pts, np.uint64, wi), self._np(scalars, np.uint8, 32)
        if len(s) % ns or len(p) * ns != len(s) * m:
            raise ValueError("need m points and %d scalars per group" % ns)
        n = len(s) // ns
        out, inf = np.empty((n, w), np.uint64), np.empty(n, np.uint8)
        f = getattr(self._lib, fname)
        args = (self._h, p.ctypes.data, in_form, s.ctypes.data) + ((m,) if "many" in fname else ()) + (out.ctypes.data, inf.ctypes.data, n, out_form)
        self._check(f(*args))
        return out, inf

    def multiply_many_batch(self, pts, scalars, m, in_form=AFFINE, out_form=AFFINE):
        """out[g] = sum_j scalars[g*m + j] * pts[g*m + j], constant time (multiply_many,
        src/curve/projective.rs:817-832, src/curve/jacobian.rs:807-823; m = 2: multiply_double)."""
        return self._many("cheetah_multiply_many_batch", pts, scalars, m, m, in_form, out_form)

    def multiply_many_vartime_batch(self, pts, scalars, m, in_form=AFFINE, out_form=AFFINE):
        """Variable-time form (multiply_many_vartime, src/curve/projective.rs:840-869; m = 2: multiply_double_vartime)."""
        return self._many("cheetah_multiply_many_vartime_batch", pts, scalars, m, m, in_form, out_form)

    def multiply_double_with_basepoint_vartime_batch(self, pts, scalars, in_form=AFFINE, out_form=AFFINE):
        """out[i] = scalars[2i] * pts[i] + scalars[2i + 1] * G (src/curve/projective.rs:756-809)."""
        return self._many("cheetah_multiply_double_with_basepoint_vartime_batch", pts, scalars, 1, 2, in_form, out_form)

    def is_torsion_free_batch(self, pts, in_form=AFFINE):
        """flags[i] = (q * pts[i]).is_identity()   (is_torsion_free, src/curve/projective.rs:885-893)"""
        wi = self._out_width(in_form)
        if _is_torch(pts):
            p = self._dev(pts, 8, wi)
            flags = self._empty((p.shape[0],), torch.uint8)
            self._use_torch_stream()
            self._check(self._lib.cheetah_is_torsion_free_batch_dev(self._h, p.data_ptr(), in_form, flags.data_ptr(), p.shape[0]))
            return flags
        p = self._np(pts, np.uint64, wi)
        flags = np.empty(len(p), np.uint8)
        self._check(self._lib.cheetah_is_torsion_free_batch(self._h, p.ctypes.data, in_form, flags.ctypes.data, len(p)))
        return flags

    def clear_cofactor_batch(self, pts, in_form=AFFINE, out_form=AFFINE):
        """out[i] = h * pts[i] through the vartime ladder (clear_cofactor, src/curve/projective.rs:872-881)."""
        w, wi = self._out_width(out_form), self._out_width(in_form)
        if _is_torch(pts):
            p = self._dev(pts, 8, wi)
            n = p.shape[0]
            out, inf = self._empty((n, w), torch.uint64), self._empty((n,), torch.uint8)
            self._use_torch_stream()
            self._check(self._lib.cheetah_clear_cofactor_batch_dev(self._h, p.data_ptr(), in_form, out.data_ptr(), inf.data_ptr(), n, out_form))
            return out, inf
        p = self._np(pts, np.uint64, wi)
        n = len(p)
        out, inf = np.empty((n, w), np.uint64), np.empty(n, np.uint8)
        self._check(self._lib.cheetah_clear_cofactor_batch(self._h, p.ctypes.data, in_form, out.ctypes.data, inf.ctypes.data, n, out_form))
        return out, inf

    # -- point encodings ------------------------------------------------------------
    def _encode(self, xy, inf, rec):
        name = "cheetah_encode_uncompressed_batch" if rec == 97 else "cheetah_encode_compressed_batch"
        if _is_torch(xy):
            p = self._dev(xy, 8, 12)
            n = p.shape[0]
            out = self._empty((n, rec), torch.uint8)
            self._use_torch_stream()
            self._check(getattr(self._lib, name + "_dev")(self._h, p.data_ptr(), inf.data_ptr(), out.data_ptr(), n))
            return out
        p = self._np(xy, np.uint64, 12)
        inf = np.ascontiguousarray(inf, np.uint8)
        out = np.empty((len(p), rec), np.uint8)
        self._check(getattr(self._lib, name)(self._h, p.ctypes.data, inf.ctypes.data, out.ctypes.data, len(p)))
        return out

    def to_uncompressed_batch(self, xy, inf):
        """97-byte encodings (AffinePoint::to_uncompressed, src/curve/encoding.rs:169-183)."""
        return self._encode(xy, inf, 97)

    def to_compressed_batch(self, xy, inf):
        """49-byte encodings (AffinePoint::to_compressed, src/curve/encoding.rs:55-77)."""
        return self._encode(xy, inf, 49)

    def from_uncompressed_batch(self, b, checked=True):
        """(xy, inf, is_some): AffinePoint::from_uncompressed (checked: on the curve and torsion free,
        src/curve/affine.rs:289-292) or from_uncompressed_unchecked (src/curve/encoding.rs:187-226)."""
        if _is_torch(b):
            b = self._dev(b, 1, 97)
            n = b.shape[0]
            xy, inf, some = self._empty((n, 12), torch.uint64), self._empty((n,), torch.uint8), self._empty((n,), torch.uint8)
            self._use_torch_stream()
            self._check(self._lib.cheetah_decode_uncompressed_batch_dev(self._h, b.data_ptr(), xy.data_ptr(), inf.data_ptr(),
                                                                        some.data_ptr(), n, int(checked)))
            return xy, inf, some
        b = self._np(b, np.uint8, 97)
        n = len(b)
        xy, inf, some = np.empty((n, 12), np.uint64), np.empty(n, np.uint8), np.empty(n, np.uint8)
        self._check(self._lib.cheetah_decode_uncompressed_batch(self._h, b.ctypes.data, xy.ctypes.data, inf.ctypes.data,
                                                                some.ctypes.data, n, int(checked)))
        return xy, inf, some

    def from_compressed_batch(self, b, checked=True):
        """(xy, inf, is_some): AffinePoint::from_compressed (checked: torsion free, src/curve/affine.rs:303-309)
        or from_compressed_unchecked = CompressedPoint::to_affine (src/curve/encoding.rs:82-129)."""
        if _is_torch(b):
            b = self._dev(b, 1, 49)
            n = b.shape[0]
            xy, inf, some = self._empty((n, 12), torch.uint64), self._empty((n,), torch.uint8), self._empty((n,), torch.uint8)
            self._use_torch_stream()
            self._check(self._lib.cheetah_decode_compressed_batch_dev(self._h, b.data_ptr(), xy.data_ptr(), inf.data_ptr(),
                                                                      some.data_ptr(), n, int(checked)))
            return xy, inf, some
        b = self._np(b, np.uint8, 49)
        n = len(b)
        xy, inf, some = np.empty((n, 12), np.uint64), np.empty(n, np.uint8), np.empty(n, np.uint8)
        self._check(self._lib.cheetah_decode_compressed_batch(self._h, b.ctypes.data, xy.ctypes.data, inf.ctypes.data,
                                                              some.ctypes.data, n, int(checked)))
        return xy, inf, some

    def from_x_batch(self, x):
        """(xy, inf, is_some): AffinePoint::from_x (src/curve/affine.rs:326-343)."""
        if _is_torch(x):
            x = self._dev(x, 8, 6)
            n = x.shape[0]
            xy, inf, some = self._empty((n, 12), torch.uint64), self._empty((n,), torch.uint8), self._empty((n,), torch.uint8)
            self._use_torch_stream()
            self._check(self._lib.cheetah_affine_from_x_batch_dev(self._h, x.data_ptr(), xy.data_ptr(), inf.data_ptr(), some.data_ptr(), n))
            return xy, inf, some
        x = self._np(x, np.uint64, 6)
        n = len(x)
        xy, inf, some = np.empty((n, 12), np.uint64), np.empty(n, np.uint8), np.empty(n, np.uint8)
        self._check(self._lib.cheetah_affine_from_x_batch(self._h, x.ctypes.data, xy.ctypes.data, inf.ctypes.data, some.ctypes.data, n))
        return xy, inf, some

    # -- tables / synthetic inputs / diagnostics -------------------------------------
    def read_table(self, which):
        n = {0: 256, 1: 3, 2: 128, 3: 64}[which]
        out = np.empty((n, 12), np.uint64)
        self._check(self._lib.cheetah_read_table(self._h, which, out.ctypes.data, n))
        return out

    def synth_fp6(self, seed, n, start=0):
        out = self._empty((n, 6), torch.uint64)
        self._use_torch_stream()
        self._check(self._lib.cheetah_synth_fp6_dev(self._h, seed, start, out.data_ptr(), n))
        return out

    def synth_scalars(self, seed, n, start=0):
        out = self._empty((n, 32), torch.uint8)
        self._use_torch_stream()
        self._check(self._lib.cheetah_synth_scalars_dev(self._h, seed, start, out.data_ptr(), n))
        return out

    def debug_op(self, op, a, b, out_width):
        a = a.contiguous()
        n = a.shape[0]
        out = self._empty((n, out_width), torch.uint64)
        self._use_torch_stream()
        self._check(self._lib.cheetah_debug_op_dev(self._h, op, a.data_ptr(), 0 if b is None else b.contiguous().data_ptr(), out.data_ptr(), n))
        return out

    def calibrate_mad_peak(self):
        v = C.c_double()
        self._use_torch_stream() if torch is not None and torch.cuda.is_available() else None
        self._check(self._lib.cheetah_calibrate_mad_peak(self._h, C.byref(v)))
        return v.value

    def set_reserved_cus(self, cus):
        """Leave `cus` CUs' worth of workgroup slots free in the persistent launches (room for an RCCL kernel on
        another stream); 0 = none."""
        self._check(self._lib.cheetah_ctx_set_reserved_cus(self._h, int(cus)))

    def set_launch_rounds(self, rounds):
        """Rounds of the resident grid one ladder / fixed-base launch covers (default 2; raise it when nothing else needs
        CU slots between launches)."""
        self._check(self._lib.cheetah_ctx_set_launch_rounds(self._h, int(rounds)))

    def set_point_independent_tables(self, on):
        """Constant-time ladders: build every wave's window tables by the reference's construction, so that timing does
        not depend on the points either (same results, 4 % slower; off by default)."""
        self._check(self._lib.cheetah_ctx_set_point_independent_tables(self._h, int(bool(on))))

    def set_reference_schedule(self, on):
        """Tests / tools: run multiply_double_with_basepoint_vartime's affine results by the reference's Straus NAF
        schedule instead of the composed one (same results)."""
        self._check(self._lib.cheetah_debug_set_reference_schedule(self._h, int(bool(on))))

    def multiply_units_per_launch(self, n, vartime=False, out_form=AFFINE):
        """Scalar multiplications per kernel launch for a batch of n (the library bounds each launch)."""
        v = C.c_size_t()
        self._check(self._lib.cheetah_multiply_units_per_launch(self._h, n, 1 if vartime else 0, out_form, C.byref(v)))
        return v.value

    def timer_start(self):
        self._check(self._lib.cheetah_timer_start(self._h))

    def timer_stop_ms(self):
        v = C.c_float()
        self._check(self._lib.cheetah_timer_stop_ms(self._h, C.byref(v)))
        return v.value


class BasePointTable:
    """BasePointTable for a caller-supplied base (src/lookup.rs:136-234): `create` at construction,
    `multiply_batch` = BasePointTable::multiply over a batch of scalars."""

    def __init__(self, engine, point, in_form=AFFINE):
        self._eng = engine
        pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1)
        if pt.size != (12 if in_form == AFFINE else 18):
            raise ValueError("one point: 12 limbs (AFFINE) or 18 (PROJECTIVE)")
        h = C.c_void_p()
        engine._check(engine._lib.cheetah_basepoint_table_create(engine._h, pt.ctypes.data, in_form, C.byref(h)))
        self._t = h

    def close(self):
        if getattr(self, "_t", None) and getattr(self._eng, "_h", None):
            self._eng._lib.cheetah_basepoint_table_destroy(self._t)
        self._t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read(self):
        """The 256 affine entries (32 x LookupTable<8>), table-major."""
        out = np.empty((256, 12), np.uint64)
        self._eng._check(self._eng._lib.cheetah_basepoint_table_read(self._eng._h, self._t, out.ctypes.data, 256))
        return out

    def multiply_batch(self, scalars, out_form=PROJECTIVE):
        e = self._eng
        w = e._out_width(out_form)
        if _is_torch(scalars):
            s = e._dev(scalars, 1, 32)
            n = s.shape[0]
            out, inf = e._empty((n, w), torch.uint64), e._empty((n,), torch.uint8)
            e._use_torch_stream()
            e._check(e._lib.cheetah_basepoint_table_multiply_batch_dev(e._h, self._t, s.data_ptr(), out.data_ptr(), inf.data_ptr(), n, out_form))
            return out, inf
        s = e._np(scalars, np.uint8, 32)
        n = len(s)
        out, inf = np.empty((n, w), np.uint64), np.empty(n, np.uint8)
        e._check(e._lib.cheetah_basepoint_table_multiply_batch(e._h, self._t, s.ctypes.data, out.ctypes.data, inf.ctypes.data, n, out_form))
        return out, inf


# ---------------------------------------------------------------------------
# Reference-shaped facade: the names a user of the crate looks for.

_default = {}


def default_engine(device=0):
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]



def _interleave(a, b):
    """[a0, b0, a1, b1, ...]: the per-group layout of the multi-scalar entry points."""
    if _is_torch(a):
        return torch.stack([a, b], dim=1).reshape(2 * a.shape[0], a.shape[1]).contiguous()
    return np.stack([a, b], axis=1).reshape(2 * len(a), a.shape[1])


class Fp6:
    @staticmethod
    def mul_batch(a, b, engine=None):
        return (engine or default_engine()).fp6_mul_batch(a, b)

    @staticmethod
    def square_batch(a, engine=None):
        return (engine or default_engine()).fp6_square_batch(a)

    @staticmethod
    def invert_batch(a, engine=None):
        return (engine or default_engine()).fp6_invert_batch(a)

    @staticmethod
    def sqrt_batch(a, engine=None):
        return (engine or default_engine()).fp6_sqrt_batch(a)


class ProjectivePoint:
    @staticmethod
    def batch_normalize(p, engine=None):
        return (engine or default_engine()).batch_normalize(p, PROJECTIVE)

    @staticmethod
    def multiply_double_batch(lhs, rhs, by_lhs, by_rhs, in_form=PROJECTIVE, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).multiply_many_batch(_interleave(lhs, rhs), _interleave(by_lhs, by_rhs), 2, in_form, out_form)

    @staticmethod
    def multiply_double_vartime_batch(lhs, rhs, by_lhs, by_rhs, in_form=PROJECTIVE, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).multiply_many_vartime_batch(_interleave(lhs, rhs), _interleave(by_lhs, by_rhs), 2, in_form, out_form)

    @staticmethod
    def multiply_double_with_basepoint_vartime_batch(points, by_self, by_basepoint, in_form=PROJECTIVE, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).multiply_double_with_basepoint_vartime_batch(points, _interleave(by_self, by_basepoint), in_form, out_form)

    @staticmethod
    def multiply_many_batch(points, scalars, m, in_form=PROJECTIVE, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).multiply_many_batch(points, scalars, m, in_form, out_form)

    @staticmethod
    def multiply_many_vartime_batch(points, scalars, m, in_form=PROJECTIVE, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).multiply_many_vartime_batch(points, scalars, m, in_form, out_form)

    @staticmethod
    def is_torsion_free_batch(points, in_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).is_torsion_free_batch(points, in_form)

    @staticmethod
    def clear_cofactor_batch(points, in_form=PROJECTIVE, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).clear_cofactor_batch(points, in_form, out_form)

    @staticmethod
    def multiply_batch(points, scalars, in_form=PROJECTIVE, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).multiply_batch(points, scalars, in_form, out_form)

    @staticmethod
    def multiply_vartime_batch(points, scalars, in_form=PROJECTIVE, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).multiply_vartime_batch(points, scalars, in_form, out_form)


class JacobianPoint:
    @staticmethod
    def batch_normalize(p, engine=None):
        return (engine or default_engine()).batch_normalize(p, JACOBIAN)

    @staticmethod
    def multiply_double_batch(lhs, rhs, by_lhs, by_rhs, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_many_batch(_interleave(lhs, rhs), _interleave(by_lhs, by_rhs), 2, in_form, out_form)

    @staticmethod
    def multiply_double_vartime_batch(lhs, rhs, by_lhs, by_rhs, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_many_vartime_batch(_interleave(lhs, rhs), _interleave(by_lhs, by_rhs), 2, in_form, out_form)

    @staticmethod
    def multiply_double_with_basepoint_vartime_batch(points, by_self, by_basepoint, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_double_with_basepoint_vartime_batch(points, _interleave(by_self, by_basepoint), in_form, out_form)

    @staticmethod
    def multiply_many_batch(points, scalars, m, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_many_batch(points, scalars, m, in_form, out_form)

    @staticmethod
    def multiply_many_vartime_batch(points, scalars, m, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_many_vartime_batch(points, scalars, m, in_form, out_form)

    @staticmethod
    def is_torsion_free_batch(points, in_form=JACOBIAN, engine=None):
        return (engine or default_engine()).is_torsion_free_batch(points, in_form)

    @staticmethod
    def clear_cofactor_batch(points, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).clear_cofactor_batch(points, in_form, out_form)

    @staticmethod
    def multiply_batch(points, scalars, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_batch(points, scalars, in_form, out_form)

    @staticmethod
    def multiply_vartime_batch(points, scalars, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_vartime_batch(points, scalars, in_form, out_form)


class AffinePoint:
    @staticmethod
    def to_uncompressed_batch(xy, inf, engine=None):
        return (engine or default_engine()).to_uncompressed_batch(xy, inf)

    @staticmethod
    def to_compressed_batch(xy, inf, engine=None):
        return (engine or default_engine()).to_compressed_batch(xy, inf)

    @staticmethod
    def from_uncompressed_batch(b, engine=None):
        return (engine or default_engine()).from_uncompressed_batch(b, True)

    @staticmethod
    def from_uncompressed_unchecked_batch(b, engine=None):
        return (engine or default_engine()).from_uncompressed_batch(b, False)

    @staticmethod
    def from_compressed_batch(b, engine=None):
        return (engine or default_engine()).from_compressed_batch(b, True)

    @staticmethod
    def from_compressed_unchecked_batch(b, engine=None):
        return (engine or default_engine()).from_compressed_batch(b, False)

    @staticmethod
    def from_x_batch(x, engine=None):
        return (engine or default_engine()).from_x_batch(x)

    @staticmethod
    def multiply_batch(points, scalars, engine=None):
        return (engine or default_engine()).multiply_batch(points, scalars, AFFINE, AFFINE)

    @staticmethod
    def multiply_vartime_batch(points, scalars, engine=None):
        return (engine or default_engine()).multiply_vartime_batch(points, scalars, AFFINE, AFFINE)


class _BasePointTable:
    """`&BASEPOINT_TABLE * &Scalar` (src/constants.rs:92-98) over a batch."""

    @staticmethod
    def multiply_batch(scalars, out_form=PROJECTIVE, engine=None):
        return (engine or default_engine()).mul_by_generator_batch(scalars, out_form)


class _BasePointLookup:
    """`&BASEPOINT_LOOKUP * &Scalar` (src/constants.rs:102-135) over a batch: Jacobian results."""

    @staticmethod
    def multiply_batch(scalars, engine=None):
        return (engine or default_engine()).mul_by_generator_batch(scalars, JACOBIAN)


BASEPOINT_TABLE = _BasePointTable()
BASEPOINT_LOOKUP = _BasePointLookup()
