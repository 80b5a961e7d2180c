//! Status codes of the C ABI mapped onto the reference's error types (`include/jolt_hip.h:44-59`).
use core::ffi::CStr;

use jolt_field::{CanonicalEncoding, Fr};
use jolt_kernels::{KernelError, SumcheckKernelError};
use jolt_sumcheck::SumcheckError;

use crate::ffi;

/// A non-zero status together with the context's `jolt_last_error` text.
#[derive(Debug, Clone, thiserror::Error)]
#[error("libjolt_hip status {status} ({name}): {detail}")]
pub struct HipError {
    pub status: i32,
    pub name: &'static str,
    pub detail: String,
}

impl HipError {
    /// `Unsupported`-class statuses are recoverable: the caller falls back to another backend's slot
    /// (`crates/jolt-kernels/src/optimized/mod.rs:136-196` composes backends slot by slot).
    /// A shape the wrapper rejects before calling into the library.
    pub fn size_mismatch(what: &'static str) -> Self {
        Self { status: ffi::JOLT_ERR_SIZE_MISMATCH, name: "JOLT_ERR_SIZE_MISMATCH", detail: what.to_owned() }
    }

    pub fn is_recoverable(&self) -> bool {
        matches!(self.status, ffi::JOLT_ERR_NO_DEVICE | ffi::JOLT_ERR_OOM | ffi::JOLT_ERR_UNSUPPORTED)
    }
}

pub(crate) fn check(status: i32, ctx: *const ffi::jolt_ctx) -> Result<(), HipError> {
    if status == ffi::JOLT_OK {
        return Ok(());
    }
    // SAFETY: jolt_status_string returns a pointer to a static NUL-terminated string for every status value.
    let name = unsafe { CStr::from_ptr(ffi::jolt_status_string(status)) }.to_str().unwrap_or("?");
    let detail = if ctx.is_null() {
        String::new()
    } else {
        // SAFETY: `ctx` is a live context (callers hold an Arc<HipContext>); the returned string is owned by it and copied here.
        unsafe { CStr::from_ptr(ffi::jolt_last_error(ctx)) }.to_string_lossy().into_owned()
    };
    Err(HipError { status, name, detail })
}

impl From<HipError> for KernelError<Fr> {
    /// `crates/jolt-kernels/src/error.rs:11-90`: capability gaps are `Unsupported` (the caller may retry the slot against another
    /// backend), everything else is a bug.  The detailed text goes to the tracing span, the variants carry `&'static str`.
    fn from(e: HipError) -> Self {
        tracing::error!(status = e.status, detail = %e.detail, "libjolt_hip call failed");
        match e.status {
            ffi::JOLT_ERR_NO_DEVICE => KernelError::Unsupported { reason: "no usable gfx950 device" },
            ffi::JOLT_ERR_OOM => KernelError::Unsupported { reason: "out of device memory" },
            ffi::JOLT_ERR_UNSUPPORTED => KernelError::Unsupported { reason: "descriptor beyond the compiled limits of libjolt_hip" },
            ffi::JOLT_ERR_SIZE_MISMATCH => KernelError::InvariantViolation { reason: "table sizes disagree with the relation's rounds" },
            _ => KernelError::InvariantViolation { reason: "libjolt_hip reported an error (see the tracing log)" },
        }
    }
}

/// `SumcheckKernel::output_claims` failures (`crates/jolt-kernels/src/kernel.rs:25-53`).
pub(crate) fn to_kernel_seam_error(e: HipError, remaining: usize) -> SumcheckKernelError<Fr> {
    tracing::error!(status = e.status, detail = %e.detail, "libjolt_hip call failed");
    match e.status {
        ffi::JOLT_ERR_NOT_FULLY_BOUND => SumcheckKernelError::NotFullyBound { remaining },
        _ => SumcheckKernelError::InvariantViolation { reason: "libjolt_hip reported an error (see the tracing log)" },
    }
}

/// `prove_round` / `finish_rounds` report through `SumcheckError` (`crates/jolt-sumcheck/src/error.rs:12-110`): a device failure
/// between construction and the round loop is the "evaluation source went missing" case of the reference tier.
pub(crate) fn to_sumcheck_error(e: HipError) -> SumcheckError<Fr> {
    tracing::error!(status = e.status, detail = %e.detail, "libjolt_hip call failed");
    SumcheckError::MissingEvaluationSource { kind: "device" }
}

/// A field element that is a machine integer, decoded (`CanonicalEncoding::to_u64_checked`, `crates/jolt-field/src/algebra.rs:300-313`).
pub(crate) fn fr_to_u64(v: &Fr) -> Option<u64> {
    v.to_u64_checked()
}
/// The same for the signed 128-bit range: a value above 2^127 is the negative of a small one (`F::from_i128`'s image).
pub(crate) fn fr_to_i128(v: &Fr) -> Option<i128> {
    v.to_u128_checked().and_then(|u| i128::try_from(u).ok()).or_else(|| (-*v).to_u128_checked().and_then(|u| i128::try_from(u).ok()).map(|m| -m))
}

/// `MaybeAllocative` for this crate's session entries and kernels.  Without the `allocative` feature the reference blanket-implements it for every type
/// (`impl<T: ?Sized> MaybeAllocative for T`, `crates/jolt-kernels/src/backend.rs:195-198`): a manual impl would CONFLICT (E0119), so nothing is written.  With the feature it
/// is `allocative::Allocative`: the state lives in HBM, the host heap behind it is a few handles -- reported as zero bytes (`optimized::impl_allocative!`'s shape).
macro_rules! zero_host_heap {
    ($type:ty) => {
        #[cfg(feature = "allocative")]
        impl allocative::Allocative for $type {
            fn visit<'a, 'b: 'a>(&self, visitor: &'a mut allocative::Visitor<'b>) {
                let mut visitor = visitor.enter_self_sized::<Self>();
                visitor.visit_simple(allocative::Key::new("heap"), 0usize);
                visitor.exit();
            }
        }
    };
}
pub(crate) use zero_host_heap;
