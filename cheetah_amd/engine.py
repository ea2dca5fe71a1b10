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
    """One context = one GPU (cheetah_ctx).  Builds BASEPOINT_TABLE and the shift
    points on the device at construction (src/constants.rs:21-22, :43-90)."""

    def __init__(self, device=0):
        self._lib = _capi.load()
        self.device = int(device)
        h = C.c_void_p()
        rc = self._lib.cheetah_ctx_create(self.device, C.byref(h))
        if rc != 0:
            raise CheetahError(rc, self._lib.cheetah_last_error(None).decode())
        self._h = h

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

    def fp6_invert_batch(self, a):
        """(out, is_some): out[i] = a[i]^-1, is_some[i] = 0 and out[i] = 0 for zero
        (Fp6::invert -> CtOption, src/fp6.rs:1035-1049)"""
        if _is_torch(a):
            a = self._dev(a, 8, 6)
            out = self._empty(a.shape, torch.uint64)
            some = self._empty((a.shape[0],), torch.uint8)
            self._use_torch_stream()
            self._check(self._lib.cheetah_fp6_invert_batch_dev(self._h, a.data_ptr(), out.data_ptr(), some.data_ptr(), a.shape[0]))
            return out, some
        a = self._np(a, np.uint64, 6)
        out = np.empty_like(a)
        some = np.empty(len(a), np.uint8)
        self._check(self._lib.cheetah_fp6_invert_batch(self._h, a.ctypes.data, out.ctypes.data, some.ctypes.data, len(a)))
        return out, some

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

    # -- tables / synthetic inputs / diagnostics -------------------------------------
    def read_table(self, which):
        n = {0: 256, 1: 3, 2: 128}[which]
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

    def timer_start(self):
        self._check(self._lib.cheetah_timer_start(self._h))

    def timer_stop_ms(self):
        v = C.c_float()
        self._check(self._lib.cheetah_timer_stop_ms(self._h, C.byref(v)))
        return v.value


# ---------------------------------------------------------------------------
# Reference-shaped facade: the names a user of the crate looks for.

_default = {}


def default_engine(device=0):
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]


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


class ProjectivePoint:
    @staticmethod
    def batch_normalize(p, engine=None):
        return (engine or default_engine()).batch_normalize(p, PROJECTIVE)

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
    def multiply_batch(points, scalars, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_batch(points, scalars, in_form, out_form)

    @staticmethod
    def multiply_vartime_batch(points, scalars, in_form=JACOBIAN, out_form=JACOBIAN, engine=None):
        return (engine or default_engine()).multiply_vartime_batch(points, scalars, in_form, out_form)


class AffinePoint:
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
