//! Owning wrappers of the opaque handles: one [`HipContext`] per GPU per process, device-resident [`HipTable`]s.
use std::ptr;
use std::sync::Arc;

use jolt_field::{Fr, Ring};

use crate::ffi;
use crate::status::{check, HipError};

/// `jolt_ctx`: device + stream + scratch.  One per GPU per process; ranks shard the hypercube (`DESIGN.md` section 6).
pub struct HipContext {
    pub(crate) raw: *mut ffi::jolt_ctx,
    /// Serialises entry points that may be reached from several host threads at once (`JoltGroup::msm` under rayon): see `exclusive`.
    device: std::sync::Mutex<()>,
}

// SAFETY: the C library serialises nothing internally, so a context is used from one thread at a time (`prove_batch` is
// single-threaded, `crates/jolt-sumcheck/src/prover.rs:124-146`); handing it to another thread between calls is sound.
unsafe impl Send for HipContext {}

impl HipContext {
    /// `JOLT_ERR_NO_DEVICE` (no gfx950) is recoverable: the caller keeps the `optimized()` slots.
    pub fn new(device_id: i32) -> Result<Arc<Self>, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: `raw` is a valid out-pointer; a null stream asks the library to create its own.
        check(unsafe { ffi::jolt_ctx_create(device_id, ptr::null_mut(), &mut raw) }, ptr::null())?;
        Ok(Arc::new(Self { raw, device: std::sync::Mutex::new(()) }))
    }

    /// The context's device lock.  `prove_batch` drives its members from one thread and never needs it; callers that ARE concurrent
    /// (the MSM drop-in, `msm.rs`) hold it across their device call.  A poisoned lock is taken over: the guarded state is `()`.
    pub(crate) fn exclusive(&self) -> std::sync::MutexGuard<'_, ()> {
        self.device.lock().unwrap_or_else(std::sync::PoisonError::into_inner)
    }

    pub fn synchronize(&self) -> Result<(), HipError> {
        // SAFETY: live context.
        check(unsafe { ffi::jolt_ctx_synchronize(self.raw) }, self.raw)
    }

    /// Cached device blocks back to the runtime (between proofs of very different sizes).
    pub fn trim(&self) -> Result<(), HipError> {
        // SAFETY: live context.
        check(unsafe { ffi::jolt_ctx_trim(self.raw) }, self.raw)
    }

    /// `witness.oracle_table(id)` materialised once and uploaded (`crates/jolt-witness/src/backend/mod.rs:50-53`).
    pub fn upload(self: &Arc<Self>, evals: &[Fr]) -> Result<HipTable, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: `Fr` is `#[repr(transparent)]` over 4 x u64 Montgomery limbs (crates/jolt-field/src/bn254/mod.rs:33-43), the layout
        // of `jolt_fr_t`; the slice outlives the (synchronous) upload.
        check(unsafe { ffi::jolt_table_upload(self.raw, evals.as_ptr().cast(), evals.len(), &mut raw) }, self.raw)?;
        Ok(HipTable { ctx: Arc::clone(self), raw })
    }

    /// Small-scalar witness column: 8 bytes per entry over PCIe, promoted on the device (`Polynomial::bind_to_field`'s `From<u64>`).
    pub fn upload_u64(self: &Arc<Self>, values: &[u64]) -> Result<HipTable, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: plain integers; the slice outlives the synchronous upload.
        check(unsafe { ffi::jolt_table_from_u64(self.raw, values.as_ptr(), values.len(), &mut raw) }, self.raw)?;
        Ok(HipTable { ctx: Arc::clone(self), raw })
    }

    /// `EqPolynomial::evals(r, scale)` built on the device (`crates/jolt-poly/src/eq.rs:221-231`).
    pub fn eq_evals(self: &Arc<Self>, r: &[Fr], scale: Option<Fr>) -> Result<HipTable, HipError> {
        let mut raw = ptr::null_mut();
        let scale_ptr = scale.as_ref().map_or(ptr::null(), |s| (s as *const Fr).cast());
        // SAFETY: layouts as above; pointers valid for the call.
        check(unsafe { ffi::jolt_eq_evals(self.raw, r.as_ptr().cast(), r.len(), scale_ptr, &mut raw) }, self.raw)?;
        Ok(HipTable { ctx: Arc::clone(self), raw })
    }

    /// `LtPolynomial::evaluations` (`crates/jolt-poly/src/lt.rs:115-117,144-156`).
    pub fn lt_evals(self: &Arc<Self>, r: &[Fr]) -> Result<HipTable, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: as above.
        check(unsafe { ffi::jolt_lt_evals(self.raw, r.as_ptr().cast(), r.len(), &mut raw) }, self.raw)?;
        Ok(HipTable { ctx: Arc::clone(self), raw })
    }
}

impl HipContext {
    /// `EqPlusOnePolynomial::evals(r, None).1` (`crates/jolt-poly/src/eq_plus_one.rs:71-130`): the eq+1 table (its eq companion is dropped).
    pub fn eq_plus_one_evals(self: &Arc<Self>, r: &[Fr]) -> Result<HipTable, HipError> {
        let (mut eq, mut eq1) = (ptr::null_mut(), ptr::null_mut());
        // SAFETY: layouts as above; both out-pointers valid.
        check(unsafe { ffi::jolt_eq_plus_one_evals(self.raw, r.as_ptr().cast(), r.len(), ptr::null(), &mut eq, &mut eq1) }, self.raw)?;
        drop(HipTable { ctx: Arc::clone(self), raw: eq });
        Ok(HipTable { ctx: Arc::clone(self), raw: eq1 })
    }

    /// `sum_i scalars[i] * tables[i]` (`jolt_rlc`; `RlcSource::to_dense`, `crates/jolt-poly/src/multilinear.rs:358-464`).
    pub fn rlc(self: &Arc<Self>, tables: &[&HipTable], scalars: &[Fr]) -> Result<HipTable, HipError> {
        if tables.len() != scalars.len() || tables.is_empty() {
            return Err(HipError::size_mismatch("one scalar per table, at least one table"));
        }
        let handles: Vec<*mut ffi::jolt_table> = tables.iter().map(|t| t.raw).collect();
        let mut raw = ptr::null_mut();
        // SAFETY: live handles of equal length; one scalar per table.
        check(unsafe { ffi::jolt_rlc(self.raw, handles.as_ptr(), handles.len(), scalars.as_ptr().cast(), &mut raw) }, self.raw)?;
        Ok(HipTable { ctx: Arc::clone(self), raw })
    }

    /// `LtPolynomial::evaluations(r)[j] + constant` (`reference/ram_val_check.rs:23-26`): the LT table and a table of ones combined on the device.
    pub fn lt_evals_plus(self: &Arc<Self>, r: &[Fr], constant: Fr) -> Result<HipTable, HipError> {
        let lt = self.lt_evals(r)?;
        let ones = self.upload_u64(&vec![1u64; 1usize << r.len()])?;
        self.rlc(&[&lt, &ones], &[Fr::from_u64(1), constant])
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        // SAFETY: created by jolt_ctx_create; every handle holds an Arc to the context, so none outlives it.
        let _ = unsafe { ffi::jolt_ctx_destroy(self.raw) };
    }
}

/// `jolt_table`: a `Polynomial<Fr>` resident in HBM.
pub struct HipTable {
    pub(crate) ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_table,
}

// SAFETY: see HipContext.
unsafe impl Send for HipTable {}

impl HipTable {
    pub fn len(&self) -> usize {
        let mut n = 0usize;
        // SAFETY: live handle, valid out-pointer.
        let _ = unsafe { ffi::jolt_table_len(self.raw, &mut n) };
        n
    }

    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }

    pub fn download(&self) -> Result<Vec<Fr>, HipError> {
        let n = self.len();
        let mut out = vec![Fr::default(); n];
        // SAFETY: `out` has room for n elements of the same layout.
        check(unsafe { ffi::jolt_table_download(self.ctx.raw, self.raw, 0, n, out.as_mut_ptr().cast()) }, self.ctx.raw)?;
        Ok(out)
    }

    /// `Polynomial::evaluate` on the device (`jolt_table_evaluate`: one eq expansion, one dot product).
    pub fn evaluate(&self, point: &[Fr]) -> Result<Fr, HipError> {
        let mut out = Fr::default();
        // SAFETY: live handles; `point` holds `point.len()` field elements, `out` one.
        check(unsafe { ffi::jolt_table_evaluate(self.ctx.raw, self.raw, point.as_ptr().cast(), point.len(), (&mut out as *mut Fr).cast()) }, self.ctx.raw)?;
        Ok(out)
    }

    /// Hand the handle to a member that takes ownership (`jolt_member_create_expr`): the table is then freed with the member.
    pub(crate) fn into_raw(self) -> *mut ffi::jolt_table {
        let raw = self.raw;
        std::mem::forget(self);
        raw
    }
}

impl Drop for HipTable {
    fn drop(&mut self) {
        // SAFETY: owned handle of a live context.
        let _ = unsafe { ffi::jolt_table_free(self.ctx.raw, self.raw) };
    }
}
