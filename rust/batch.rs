//! Batch operations on an MI355X through libcheetah_hip.so (feature `hip`).
//!
//! Drop-in file for the cheetah crate: `src/batch.rs`, declared in `src/lib.rs` as
//! `#[cfg(feature = "hip")] #[allow(unsafe_code)] pub mod batch;`.  Every function is the batch form of the
//! method named in its doc comment and returns, element by element, what that method returns (canonical limbs).
//! Marshalling goes through the crate's public accessors only -- nothing is transmuted (struct layout is
//! repr(Rust) and AffinePoint.infinity is a subtle::Choice).
//!
//! There is no CPU fallback: without a gfx950 device `GpuContext::new` fails and the caller keeps using the
//! crate's own one-operation-per-call API.

use alloc::string::String;
use alloc::vec;
use alloc::vec::Vec;
use core::ffi::CStr;

use cheetah_hip_sys as sys;

use crate::{AffinePoint, Fp6, JacobianPoint, ProjectivePoint, Scalar};

/// A failed call: the C status and the library's message.
#[derive(Debug, Clone)]
pub struct BatchError {
    pub status: i32,
    pub message: String,
}

/// Device tables (BASEPOINT_TABLE, BASEPOINT_LOOKUP, shift points) + a stream, on one GPU or several.
/// `Send`, not `Sync`: one host thread at a time.
pub struct GpuContext(*mut sys::cheetah_ctx);
unsafe impl Send for GpuContext {}

impl GpuContext {
    /// A context on one device.
    pub fn new(device: i32) -> Result<Self, BatchError> {
        Self::with_devices(&[device])
    }

    /// One context over several GPUs of the node: every batch call is sharded contiguously over them and the
    /// results are gathered into the caller's slices.
    pub fn with_devices(devices: &[i32]) -> Result<Self, BatchError> {
        let mut h: *mut sys::cheetah_ctx = core::ptr::null_mut();
        let rc = unsafe { sys::cheetah_ctx_create(devices.as_ptr(), devices.len() as i32, &mut h) };
        if rc != sys::CHEETAH_OK {
            return Err(error_of(core::ptr::null(), rc));
        }
        Ok(GpuContext(h))
    }

    pub fn num_devices(&self) -> i32 {
        unsafe { sys::cheetah_ctx_num_devices(self.0) }
    }

    fn check(&self, rc: i32) -> Result<(), BatchError> {
        if rc == sys::CHEETAH_OK {
            Ok(())
        } else {
            Err(error_of(self.0, rc))
        }
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe {
            sys::cheetah_ctx_destroy(self.0);
        }
    }
}

fn error_of(ctx: *const sys::cheetah_ctx, status: i32) -> BatchError {
    let p = unsafe { sys::cheetah_last_error(ctx) };
    let message = if p.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    };
    BatchError { status, message }
}

// ---- marshalling ------------------------------------------------------------------------------------------

fn fp6_limbs(v: &[Fp6]) -> Vec<[u64; 6]> {
    v.iter().map(Fp6::output_internal).collect()
}

fn fp6_from(v: Vec<[u64; 6]>) -> Vec<Fp6> {
    v.into_iter().map(Fp6::from_raw_unchecked).collect()
}

fn cat2(a: [u64; 6], b: [u64; 6]) -> [u64; 12] {
    let mut o = [0u64; 12];
    o[..6].copy_from_slice(&a);
    o[6..].copy_from_slice(&b);
    o
}

fn cat3(a: [u64; 6], b: [u64; 6], c: [u64; 6]) -> [u64; 18] {
    let mut o = [0u64; 18];
    o[..6].copy_from_slice(&a);
    o[6..12].copy_from_slice(&b);
    o[12..].copy_from_slice(&c);
    o
}

fn six(s: &[u64]) -> Fp6 {
    let mut l = [0u64; 6];
    l.copy_from_slice(s);
    Fp6::from_raw_unchecked(l)
}

/// x | y; the identity travels as (0, 1), which is what AffinePoint::identity() holds (src/curve/affine.rs:224-230).
fn affine_limbs(p: &[AffinePoint]) -> Vec<[u64; 12]> {
    p.iter()
        .map(|p| cat2(p.get_x().output_internal(), p.get_y().output_internal()))
        .collect()
}

fn affine_from(xy: &[[u64; 12]], inf: &[u8]) -> Vec<AffinePoint> {
    xy.iter()
        .zip(inf)
        .map(|(l, &i)| {
            if i != 0 {
                AffinePoint::identity()
            } else {
                AffinePoint::from_raw_coordinates([six(&l[..6]), six(&l[6..])])
            }
        })
        .collect()
}

fn projective_limbs(p: &[ProjectivePoint]) -> Vec<[u64; 18]> {
    p.iter()
        .map(|p| cat3(p.get_x().output_internal(), p.get_y().output_internal(), p.get_z().output_internal()))
        .collect()
}

fn jacobian_limbs(p: &[JacobianPoint]) -> Vec<[u64; 18]> {
    p.iter()
        .map(|p| cat3(p.get_x().output_internal(), p.get_y().output_internal(), p.get_z().output_internal()))
        .collect()
}

fn scalar_bytes(k: &[Scalar]) -> Vec<[u8; 32]> {
    k.iter().map(Scalar::to_bytes).collect()
}

// ---- Fp6 --------------------------------------------------------------------------------------------------

impl Fp6 {
    /// out[i] = a[i] * b[i]  (Fp6::mul).  Panics on a length mismatch, like the crate's slice APIs.
    pub fn mul_batch(ctx: &mut GpuContext, a: &[Fp6], b: &[Fp6]) -> Result<Vec<Fp6>, BatchError> {
        assert_eq!(a.len(), b.len());
        let (fa, fb) = (fp6_limbs(a), fp6_limbs(b));
        let mut out = vec![[0u64; 6]; a.len()];
        let rc = unsafe {
            sys::cheetah_fp6_mul_batch(ctx.0, fa.as_ptr().cast(), fb.as_ptr().cast(), out.as_mut_ptr().cast(), a.len())
        };
        ctx.check(rc)?;
        Ok(fp6_from(out))
    }

    /// out[i] = a[i]^2  (Fp6::square).
    pub fn square_batch(ctx: &mut GpuContext, a: &[Fp6]) -> Result<Vec<Fp6>, BatchError> {
        let fa = fp6_limbs(a);
        let mut out = vec![[0u64; 6]; a.len()];
        let rc = unsafe { sys::cheetah_fp6_square_batch(ctx.0, fa.as_ptr().cast(), out.as_mut_ptr().cast(), a.len()) };
        ctx.check(rc)?;
        Ok(fp6_from(out))
    }

    /// out[i] = a[i]^-1, None for zero  (Fp6::invert, a CtOption there).
    pub fn invert_batch(ctx: &mut GpuContext, a: &[Fp6]) -> Result<Vec<Option<Fp6>>, BatchError> {
        let fa = fp6_limbs(a);
        let mut out = vec![[0u64; 6]; a.len()];
        let mut some = vec![0u8; a.len()];
        let rc = unsafe {
            sys::cheetah_fp6_invert_batch(ctx.0, fa.as_ptr().cast(), out.as_mut_ptr().cast(), some.as_mut_ptr(), a.len())
        };
        ctx.check(rc)?;
        Ok(out
            .into_iter()
            .zip(some)
            .map(|(l, s)| if s != 0 { Some(Fp6::from_raw_unchecked(l)) } else { None })
            .collect())
    }
}

impl Fp6 {
    /// out[i] = a[i]^power, power as six little-endian u64 words  (Fp6::exp / exp_vartime).
    pub fn exp_batch(ctx: &mut GpuContext, a: &[Fp6], power: &[u64; 6]) -> Result<Vec<Fp6>, BatchError> {
        let fa = fp6_limbs(a);
        let mut out = vec![[0u64; 6]; a.len()];
        let rc = unsafe { sys::cheetah_fp6_exp_batch(ctx.0, fa.as_ptr().cast(), power.as_ptr(), out.as_mut_ptr().cast(), a.len()) };
        ctx.check(rc)?;
        Ok(fp6_from(out))
    }
}

// ---- variable base ----------------------------------------------------------------------------------------

fn multiply_affine(ctx: &mut GpuContext, p: &[AffinePoint], k: &[Scalar], vartime: bool) -> Result<Vec<AffinePoint>, BatchError> {
    assert_eq!(p.len(), k.len());
    let (pl, kb) = (affine_limbs(p), scalar_bytes(k));
    let mut xy = vec![[0u64; 12]; p.len()];
    let mut inf = vec![0u8; p.len()];
    let f = if vartime { sys::cheetah_multiply_vartime_batch } else { sys::cheetah_multiply_batch };
    let rc = unsafe {
        f(ctx.0, pl.as_ptr().cast(), sys::CHEETAH_AFFINE, kb.as_ptr().cast(), xy.as_mut_ptr().cast(), inf.as_mut_ptr(), p.len(),
          sys::CHEETAH_AFFINE)
    };
    ctx.check(rc)?;
    Ok(affine_from(&xy, &inf))
}

impl AffinePoint {
    /// out[i] = scalars[i] * points[i], constant time in the scalars  (AffinePoint::multiply).
    pub fn multiply_batch(ctx: &mut GpuContext, points: &[AffinePoint], scalars: &[Scalar]) -> Result<Vec<AffinePoint>, BatchError> {
        multiply_affine(ctx, points, scalars, false)
    }

    /// Variable-time form  (AffinePoint::multiply_vartime).
    pub fn multiply_vartime_batch(ctx: &mut GpuContext, points: &[AffinePoint], scalars: &[Scalar]) -> Result<Vec<AffinePoint>, BatchError> {
        multiply_affine(ctx, points, scalars, true)
    }

    /// out[i] = scalars[i] * G  (&BASEPOINT_TABLE * &Scalar, as an affine point).
    pub fn mul_by_generator_batch(ctx: &mut GpuContext, scalars: &[Scalar]) -> Result<Vec<AffinePoint>, BatchError> {
        let kb = scalar_bytes(scalars);
        let mut xy = vec![[0u64; 12]; kb.len()];
        let mut inf = vec![0u8; kb.len()];
        let rc = unsafe {
            sys::cheetah_mul_by_generator_batch(ctx.0, kb.as_ptr().cast(), xy.as_mut_ptr().cast(), inf.as_mut_ptr(), kb.len(),
                                                sys::CHEETAH_AFFINE)
        };
        ctx.check(rc)?;
        Ok(affine_from(&xy, &inf))
    }
}

impl ProjectivePoint {
    /// out[i] = scalars[i] * points[i] as the reference's own (X : Y : Z)  (ProjectivePoint::multiply).
    pub fn multiply_batch(ctx: &mut GpuContext, points: &[ProjectivePoint], scalars: &[Scalar]) -> Result<Vec<ProjectivePoint>, BatchError> {
        assert_eq!(points.len(), scalars.len());
        let (pl, kb) = (projective_limbs(points), scalar_bytes(scalars));
        let mut out = vec![[0u64; 18]; pl.len()];
        let rc = unsafe {
            sys::cheetah_multiply_batch(ctx.0, pl.as_ptr().cast(), sys::CHEETAH_PROJECTIVE, kb.as_ptr().cast(), out.as_mut_ptr().cast(),
                                        core::ptr::null_mut(), pl.len(), sys::CHEETAH_PROJECTIVE)
        };
        ctx.check(rc)?;
        Ok(out.iter().map(|l| ProjectivePoint::from_raw_coordinates([six(&l[..6]), six(&l[6..12]), six(&l[12..])])).collect())
    }

    /// Variable-time form, the reference's own (X : Y : Z) as well  (ProjectivePoint::multiply_vartime).
    pub fn multiply_vartime_batch(ctx: &mut GpuContext, points: &[ProjectivePoint], scalars: &[Scalar]) -> Result<Vec<ProjectivePoint>, BatchError> {
        assert_eq!(points.len(), scalars.len());
        let (pl, kb) = (projective_limbs(points), scalar_bytes(scalars));
        let mut out = vec![[0u64; 18]; pl.len()];
        let rc = unsafe {
            sys::cheetah_multiply_vartime_batch(ctx.0, pl.as_ptr().cast(), sys::CHEETAH_PROJECTIVE, kb.as_ptr().cast(),
                                                out.as_mut_ptr().cast(), core::ptr::null_mut(), pl.len(), sys::CHEETAH_PROJECTIVE)
        };
        ctx.check(rc)?;
        Ok(out.iter().map(|l| ProjectivePoint::from_raw_coordinates([six(&l[..6]), six(&l[6..12]), six(&l[12..])])).collect())
    }

    /// out[i] = by_self[i] * points[i] + by_basepoint[i] * G as affine points: the verification equation of a batch of
    /// Schnorr signatures  (ProjectivePoint::multiply_double_with_basepoint_vartime, then AffinePoint::from).
    pub fn multiply_double_with_basepoint_vartime_batch(ctx: &mut GpuContext, points: &[AffinePoint], by_self: &[Scalar],
                                                        by_basepoint: &[Scalar]) -> Result<Vec<AffinePoint>, BatchError> {
        assert_eq!(points.len(), by_self.len());
        assert_eq!(points.len(), by_basepoint.len());
        let pl = affine_limbs(points);
        let mut kb = Vec::with_capacity(2 * points.len());
        for (a, b) in by_self.iter().zip(by_basepoint) {
            kb.push(a.to_bytes());
            kb.push(b.to_bytes());
        }
        let mut xy = vec![[0u64; 12]; pl.len()];
        let mut inf = vec![0u8; pl.len()];
        let rc = unsafe {
            sys::cheetah_multiply_double_with_basepoint_vartime_batch(ctx.0, pl.as_ptr().cast(), sys::CHEETAH_AFFINE, kb.as_ptr().cast(),
                                                                      xy.as_mut_ptr().cast(), inf.as_mut_ptr(), pl.len(), sys::CHEETAH_AFFINE)
        };
        ctx.check(rc)?;
        Ok(affine_from(&xy, &inf))
    }

    /// n points shaped like ProjectivePoint::random (x drawn, y = +-sqrt(x^3 + x + B), cofactor cleared), from a seed
    /// instead of an rng: element i is a function of (seed, start + i) only.
    pub fn random_batch(ctx: &mut GpuContext, seed: u64, start: u64, n: usize) -> Result<Vec<ProjectivePoint>, BatchError> {
        let mut out = vec![[0u64; 18]; n];
        let rc = unsafe {
            sys::cheetah_random_points(ctx.0, seed, start, out.as_mut_ptr().cast(), core::ptr::null_mut(), n, sys::CHEETAH_PROJECTIVE)
        };
        ctx.check(rc)?;
        Ok(out.iter().map(|l| ProjectivePoint::from_raw_coordinates([six(&l[..6]), six(&l[6..12]), six(&l[12..])])).collect())
    }

    /// q[i] = affine form of p[i]  (ProjectivePoint::batch_normalize); panics on a length mismatch as that one does.
    pub fn batch_normalize_gpu(ctx: &mut GpuContext, p: &[ProjectivePoint], q: &mut [AffinePoint]) -> Result<(), BatchError> {
        assert_eq!(p.len(), q.len());
        let pl = projective_limbs(p);
        let mut xy = vec![[0u64; 12]; pl.len()];
        let mut inf = vec![0u8; pl.len()];
        let rc = unsafe {
            sys::cheetah_batch_normalize_projective(ctx.0, pl.as_ptr().cast(), xy.as_mut_ptr().cast(), inf.as_mut_ptr(), pl.len())
        };
        ctx.check(rc)?;
        q.copy_from_slice(&affine_from(&xy, &inf));
        Ok(())
    }
}

impl JacobianPoint {
    /// out[i] = scalars[i] * points[i] as the reference's own (X, Y, Z)  (JacobianPoint::multiply).
    pub fn multiply_batch(ctx: &mut GpuContext, points: &[JacobianPoint], scalars: &[Scalar]) -> Result<Vec<JacobianPoint>, BatchError> {
        assert_eq!(points.len(), scalars.len());
        let (pl, kb) = (jacobian_limbs(points), scalar_bytes(scalars));
        let mut out = vec![[0u64; 18]; pl.len()];
        let rc = unsafe {
            sys::cheetah_multiply_batch(ctx.0, pl.as_ptr().cast(), sys::CHEETAH_JACOBIAN, kb.as_ptr().cast(), out.as_mut_ptr().cast(),
                                        core::ptr::null_mut(), pl.len(), sys::CHEETAH_JACOBIAN)
        };
        ctx.check(rc)?;
        Ok(out.iter().map(|l| JacobianPoint::from_raw_coordinates([six(&l[..6]), six(&l[6..12]), six(&l[12..])])).collect())
    }

    /// q[i] = affine form of p[i]  (JacobianPoint::batch_normalize).
    pub fn batch_normalize_gpu(ctx: &mut GpuContext, p: &[JacobianPoint], q: &mut [AffinePoint]) -> Result<(), BatchError> {
        assert_eq!(p.len(), q.len());
        let pl = jacobian_limbs(p);
        let mut xy = vec![[0u64; 12]; pl.len()];
        let mut inf = vec![0u8; pl.len()];
        let rc = unsafe {
            sys::cheetah_batch_normalize_jacobian(ctx.0, pl.as_ptr().cast(), xy.as_mut_ptr().cast(), inf.as_mut_ptr(), pl.len())
        };
        ctx.check(rc)?;
        q.copy_from_slice(&affine_from(&xy, &inf));
        Ok(())
    }
}

// ---- fixed base, caller-supplied point ----------------------------------------------------------------------

/// BasePointTable::create(&P) on the device(s) of a context; `multiply_batch` is BasePointTable::multiply over a batch.
pub struct GpuBasePointTable(*mut sys::cheetah_basepoint_table);

impl GpuBasePointTable {
    pub fn create(ctx: &mut GpuContext, point: &AffinePoint) -> Result<Self, BatchError> {
        let l = cat2(point.get_x().output_internal(), point.get_y().output_internal());
        let mut t: *mut sys::cheetah_basepoint_table = core::ptr::null_mut();
        let rc = unsafe { sys::cheetah_basepoint_table_create(ctx.0, l.as_ptr(), sys::CHEETAH_AFFINE, &mut t) };
        ctx.check(rc)?;
        Ok(GpuBasePointTable(t))
    }

    pub fn multiply_batch(&self, ctx: &mut GpuContext, scalars: &[Scalar]) -> Result<Vec<AffinePoint>, BatchError> {
        let kb = scalar_bytes(scalars);
        let mut xy = vec![[0u64; 12]; kb.len()];
        let mut inf = vec![0u8; kb.len()];
        let rc = unsafe {
            sys::cheetah_basepoint_table_multiply_batch(ctx.0, self.0, kb.as_ptr().cast(), xy.as_mut_ptr().cast(), inf.as_mut_ptr(),
                                                        kb.len(), sys::CHEETAH_AFFINE)
        };
        ctx.check(rc)?;
        Ok(affine_from(&xy, &inf))
    }
}

impl Drop for GpuBasePointTable {
    fn drop(&mut self) {
        unsafe {
            sys::cheetah_basepoint_table_destroy(self.0);
        }
    }
}
