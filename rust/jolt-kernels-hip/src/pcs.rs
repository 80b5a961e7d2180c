//! `CommitmentScheme` / `AdditivelyHomomorphic` for HyperKZG over the device (`crates/jolt-openings/src/schemes.rs:43-163`; the impl this
//! shadows: `crates/jolt-hyperkzg/src/scheme.rs:275-353`).
//!
//! Why a scheme of its own and not only the `JoltGroup::msm` seam of [`crate::msm`]: through `msm` the reference's `open`
//! (`scheme.rs:122-158`) folds the polynomial on the host and hands every level to `msm` as a fresh host slice -- 2^26 x 32 B of scalars
//! uploaded per MSM -- and the device's opening (folds, Horner passes, the RLC, the quotient scans and the paired witness commitments at
//! r / -r over one digit sort, `DESIGN.md` section 3.6) is unreachable.  [`HipHyperKzg`] routes `commit` / `open` to
//! `jolt_host_hyperkzg_commit` / `jolt_host_hyperkzg_open_with_transcript` with the polynomial RESIDENT in HBM ([`HipPoly`]); a polynomial
//! that lives on the host is uploaded once per call (`to_dense`).  `Output`, `Proof`, `VerifierSetup` and `verify` are the reference
//! scheme's own types and code: proofs are interchangeable byte for byte.
//!
//! Reference-side change this needs (one accessor; the field is `pub(crate)`, `crates/jolt-hyperkzg/src/types.rs:105-108`):
//! `impl<P: PairingGroup> HyperKZGProverSetup<P> { pub fn g1_powers(&self) -> &[P::G1] { &self.g1_powers } }`.
use std::borrow::Cow;
use std::ptr;
use std::sync::{Arc, Mutex, OnceLock};

use jolt_crypto::{Bn254, Bn254G1, Commitment, JoltGroup};
use jolt_field::Fr;
use jolt_hyperkzg::{HyperKZGCommitment, HyperKZGProof, HyperKZGProverSetup, HyperKZGScheme, HyperKZGVerifierSetup};
use jolt_openings::{AdditivelyHomomorphic, CommitmentScheme, OpeningsError};
use jolt_poly::MultilinearPoly;
use jolt_transcript::Transcript;

use crate::context::{HipContext, HipTable};
use crate::ffi;
use crate::msm::HipSrs;
use crate::status::{check, HipError};

/// The reference's prover setup plus its device copy: affine bases in HBM and the fixed-base window tables (built once, here).
/// `Clone + Send + Sync` as `CommitmentScheme::ProverSetup` demands: the device handles sit behind an `Arc<Mutex<..>>`, held across a
/// whole device call (one context, one thread at a time).
#[derive(Clone)]
pub struct HipHyperKzgSetup {
    pub inner: HyperKZGProverSetup<Bn254>,
    device: Arc<DeviceSetup>,
}

struct DeviceSetup {
    ctx: Arc<HipContext>,
    srs: Mutex<HipSrs>,
}
// SAFETY: every use of `ctx` / `srs` happens under the `srs` mutex (see `with_device`); HipContext and HipSrs are `Send`.
unsafe impl Sync for DeviceSetup {}

impl HipHyperKzgSetup {
    /// Uploads `inner.g1_powers()` once and precomputes the window tables (44 GiB for 2^26 bases: sized for 288 GB of HBM).
    pub fn new(ctx: &Arc<HipContext>, inner: HyperKZGProverSetup<Bn254>) -> Result<Self, HipError> {
        let mut srs = HipSrs::upload(ctx, inner.g1_powers())?;
        srs.precompute_windows()?;
        Ok(Self { inner, device: Arc::new(DeviceSetup { ctx: Arc::clone(ctx), srs: Mutex::new(srs) }) })
    }

    pub(crate) fn with_device<R>(&self, f: impl FnOnce(&Arc<HipContext>, &HipSrs) -> Result<R, HipError>) -> Result<R, HipError> {
        let _device = self.device.ctx.exclusive();
        let srs = self.device.srs.lock().unwrap_or_else(std::sync::PoisonError::into_inner);
        f(&self.device.ctx, &srs)
    }

    pub fn context(&self) -> &Arc<HipContext> {
        &self.device.ctx
    }

    /// One-hot columns of the commitment grid against this setup's bases: sums of selected bases (`HipHotIndices::grid_commit`).
    pub fn grid_commit_onehot(&self, columns: &crate::ops::HipHotIndices) -> Result<Vec<HyperKZGCommitment<Bn254>>, HipError> {
        self.with_device(|_, srs| Ok(columns.grid_commit(srs)?.into_iter().map(|point| HyperKZGCommitment { point }).collect()))
    }
}

/// A multilinear polynomial whose evaluation table lives in HBM (the joint polynomial `jolt_grid_joint_polynomial` writes, a witness
/// column promoted on the device).  As a `MultilinearPoly` it behaves like the dense table it holds -- `to_dense` downloads it once --
/// and [`HipHyperKzg`] recognises it by [`HipPoly::resident`] and never moves it.
pub struct HipPoly {
    table: Mutex<HipTable>,
    num_vars: usize,
    host: OnceLock<Vec<Fr>>,
}
// SAFETY: the table is only touched under its mutex; HipTable is `Send`.
unsafe impl Sync for HipPoly {}

impl HipPoly {
    pub fn new(table: HipTable) -> Result<Self, HipError> {
        let len = table.len();
        if !len.is_power_of_two() {
            return Err(HipError::size_mismatch("a multilinear polynomial has 2^n evaluations"));
        }
        Ok(Self { table: Mutex::new(table), num_vars: len.trailing_zeros() as usize, host: OnceLock::new() })
    }

    fn host(&self) -> &[Fr] {
        self.host.get_or_init(|| {
            let t = self.table.lock().unwrap_or_else(std::sync::PoisonError::into_inner);
            t.download().unwrap_or_default()
        })
    }

    /// The device table, for the entry points that take it in place.
    pub fn resident<R>(&self, f: impl FnOnce(&HipTable) -> R) -> R {
        f(&self.table.lock().unwrap_or_else(std::sync::PoisonError::into_inner))
    }
}

impl MultilinearPoly<Fr> for HipPoly {
    fn num_vars(&self) -> usize {
        self.num_vars
    }
    fn evaluate(&self, point: &[Fr]) -> Fr {
        // Polynomial::evaluate on the device (jolt_table_evaluate: one eq expansion, one dot product); host table if the call fails
        self.resident(|t| t.evaluate(point)).unwrap_or_else(|_| jolt_poly::Polynomial::new(self.host().to_vec()).evaluate(point))
    }
    fn for_each_row(&self, sigma: usize, f: &mut dyn FnMut(usize, &[Fr])) {
        for (i, row) in self.host().chunks(1usize << sigma).enumerate() {
            f(i, row);
        }
    }
    fn dense_evaluations(&self) -> Option<&[Fr]> {
        Some(self.host())
    }
    fn to_dense(&self) -> Cow<'_, [Fr]> {
        Cow::Borrowed(self.host())
    }
}

/// HyperKZG with the prover's T-scale work on the MI355X.  Verifier-side items are the reference scheme's.
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct HipHyperKzg;

impl Commitment for HipHyperKzg {
    type Output = HyperKZGCommitment<Bn254>;
}

/// What `open`'s transcript hook sees through the C ABI: the transcript and the first error it hit.
struct Hook<'a, T: Transcript<Challenge = Fr>> {
    transcript: &'a mut T,
}

/// `jolt_open_transcript_fn`: absorb what the prover sends at this step (level commitments / evaluations / witness commitments, in the
/// order `HyperKZGScheme::open` and `kzg_open_batch` append them, `scheme.rs:148-152`, `kzg.rs:88-95,118-124`) and draw the challenge.
unsafe extern "C" fn open_hook<T: Transcript<Challenge = Fr>>(
    user: *mut core::ffi::c_void,
    _phase: i32,
    points: *const ffi::jolt_g1_t,
    n_points: usize,
    values: *const ffi::jolt_fr_t,
    n_values: usize,
    challenge_out: *mut ffi::jolt_fr_t,
) -> i32 {
    // The transcript is the caller's code: a panic inside `append` / `challenge` must not unwind through the C frames of libjolt_hip.so (undefined behaviour).
    // It is caught here and reported as a status, which aborts the opening with that status (`jolt_open_transcript_fn`: "a non-zero return aborts").
    let outcome = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| {
        // SAFETY: `user` is the `Hook` `open` passes for the duration of the call; the arrays hold the stated counts of 96- / 32-byte
        // elements with the layouts of `Bn254G1` / `Fr` (crates/jolt-crypto/src/ec/bn254/mod.rs:17-24, crates/jolt-field/src/bn254/mod.rs:33-43).
        let hook = unsafe { &mut *user.cast::<Hook<'_, T>>() };
        if n_points != 0 {
            // SAFETY: as above: `n_points` points at `points`.
            for p in unsafe { std::slice::from_raw_parts(points.cast::<Bn254G1>(), n_points) } {
                hook.transcript.append(p);
            }
        }
        if n_values != 0 {
            // SAFETY: as above: `n_values` field elements at `values`.
            for v in unsafe { std::slice::from_raw_parts(values.cast::<Fr>(), n_values) } {
                hook.transcript.append(v);
            }
        }
        let challenge: Fr = hook.transcript.challenge();
        // SAFETY: `challenge_out` is one writable jolt_fr_t.
        unsafe { challenge_out.cast::<Fr>().write(challenge) };
    }));
    match outcome {
        Ok(()) => ffi::JOLT_OK,
        Err(_) => ffi::JOLT_ERR_INVALID_ARG,
    }
}

impl HipHyperKzg {
    /// `poly` as a device table: in place when it is a [`HipPoly`] (recognised through `dense_evaluations` pointing into its host cache is
    /// NOT attempted: the caller that holds a `HipPoly` uses [`HipHyperKzg::commit_resident`] / [`open_resident`] directly), uploaded
    /// once otherwise.
    fn upload<P: MultilinearPoly<Fr> + ?Sized>(ctx: &Arc<HipContext>, poly: &P) -> Result<HipTable, HipError> {
        ctx.upload(&poly.to_dense())
    }

    /// `commit` over a polynomial already in HBM.
    pub fn commit_resident(poly: &HipPoly, setup: &HipHyperKzgSetup) -> Result<HyperKZGCommitment<Bn254>, HipError> {
        setup.with_device(|ctx, srs| poly.resident(|t| Self::commit_table(ctx, srs, t)))
    }

    fn commit_table(ctx: &Arc<HipContext>, srs: &HipSrs, table: &HipTable) -> Result<HyperKZGCommitment<Bn254>, HipError> {
        let mut point = Bn254G1::default();
        // SAFETY: live handles; `point` is one jolt_g1_t.
        check(unsafe { ffi::jolt_host_hyperkzg_commit(ctx.raw, srs.raw, table.raw, (&mut point as *mut Bn254G1).cast()) }, ctx.raw)?;
        Ok(HyperKZGCommitment { point })
    }

    /// The commitments of up to three resident dense polynomials IN FLIGHT on the side lanes while `between` runs on the main stream (the one-hot columns'
    /// `jolt_grid_commit_onehot`): `jolt_msm_g1_tables_begin` / `_finish`.  The short MSMs of 64-bit columns are bound by the latency of their sort and reduction chains,
    /// the one-hot sums of bases by multiplications, so the two overlap almost entirely (the commit leg).  Falls back to one `commit` after the
    /// other -- `between` first -- when the context cannot hold them in flight (`JOLT_ERR_UNSUPPORTED`).
    pub fn commit_tables_overlapped<R>(
        tables: &[&HipTable],
        setup: &HipHyperKzgSetup,
        between: impl FnOnce(&Arc<HipContext>, &HipSrs) -> Result<R, HipError>,
    ) -> Result<(Vec<HyperKZGCommitment<Bn254>>, R), HipError> {
        setup.with_device(|ctx, srs| {
            let raws: Vec<*const ffi::jolt_table> = tables.iter().map(|t| t.raw.cast_const()).collect();
            let lens: Vec<usize> = tables.iter().map(|t| t.len()).collect();
            let mut pending: *mut ffi::jolt_msm_pending = std::ptr::null_mut();
            // SAFETY: `raws` / `lens` hold `tables.len()` live handles / lengths of this context; valid out-pointer.
            let begun = unsafe { ffi::jolt_msm_g1_tables_begin(ctx.raw, srs.raw, raws.as_ptr(), lens.as_ptr(), raws.len(), &mut pending) };
            if begun == ffi::JOLT_ERR_UNSUPPORTED || tables.is_empty() {
                let r = between(ctx, srs)?;
                let coms = tables.iter().map(|t| Self::commit_table(ctx, srs, t)).collect::<Result<Vec<_>, _>>()?;
                return Ok((coms, r));
            }
            check(begun, ctx.raw)?;
            let r = between(ctx, srs);
            let mut points = vec![Bn254G1::default(); tables.len()];
            // SAFETY: `pending` came from the begin above and is consumed exactly once, whatever `between` returned; `points` holds one jolt_g1_t per MSM.
            let finished = unsafe { ffi::jolt_msm_g1_tables_finish(ctx.raw, pending, points.as_mut_ptr().cast()) };
            let r = r?;
            check(finished, ctx.raw)?;
            Ok((points.into_iter().map(|point| HyperKZGCommitment { point }).collect(), r))
        })
    }

    /// `open` over a polynomial already in HBM: the `open` leg of the bench (`DESIGN.md` section 4.1).
    pub fn open_resident<T: Transcript<Challenge = Fr>>(
        poly: &HipPoly,
        point: &[Fr],
        setup: &HipHyperKzgSetup,
        transcript: &mut T,
    ) -> Result<HyperKZGProof<Bn254>, HipError> {
        setup.with_device(|ctx, srs| poly.resident(|t| Self::open_table(ctx, srs, t, point, transcript, &[])))
    }

    /// The same with the commitments of the first folded polynomials SUPPLIED (`jolt_host_hyperkzg_open_with_levels`): for the joint polynomial of one-hot and dense
    /// columns they follow by linearity from [`crate::ops::HipHotIndices::grid_commit_classes`] -- `com(P_s) = sum_p s_p sum_c w_c S_p^(s,c) + com(dense fold)`,
    /// `docs/kernels.md` section 3.7b; `HipHyperKzg::open_joint` (section 3.7c) is the path that computes them on the device -- instead of from MSMs over
    /// 2^(ell - s) full-width scalars.  They are absorbed and returned like computed ones.
    pub fn open_resident_with_levels<T: Transcript<Challenge = Fr>>(
        poly: &HipPoly,
        point: &[Fr],
        setup: &HipHyperKzgSetup,
        transcript: &mut T,
        known_levels: &[HyperKZGCommitment<Bn254>],
    ) -> Result<HyperKZGProof<Bn254>, HipError> {
        let points: Vec<Bn254G1> = known_levels.iter().map(|c| c.point).collect();
        setup.with_device(|ctx, srs| poly.resident(|t| Self::open_table(ctx, srs, t, point, transcript, &points)))
    }

    fn open_table<T: Transcript<Challenge = Fr>>(
        ctx: &Arc<HipContext>,
        srs: &HipSrs,
        table: &HipTable,
        point: &[Fr],
        transcript: &mut T,
        known_levels: &[Bn254G1],
    ) -> Result<HyperKZGProof<Bn254>, HipError> {
        let ell = point.len();
        let mut com = vec![Bn254G1::default(); ell.saturating_sub(1).max(1)];
        let mut w = [Bn254G1::default(); 3];
        let mut v = vec![Fr::default(); 3 * ell.max(1)];
        let mut hook = Hook { transcript };
        // SAFETY: live handles; output arrays sized as the header states (ell - 1 points, 3 points, 3 * ell field elements); the hook and
        // its transcript outlive the call; the library never unwinds.
        check(
            unsafe {
                ffi::jolt_host_hyperkzg_open_with_levels(
                    ctx.raw,
                    srs.raw,
                    table.raw,
                    point.as_ptr().cast(),
                    ell,
                    0,
                    Some(open_hook::<T>),
                    (&mut hook as *mut Hook<'_, T>).cast(),
                    if known_levels.is_empty() { ptr::null() } else { known_levels.as_ptr().cast() },
                    known_levels.len(),
                    com.as_mut_ptr().cast(),
                    w.as_mut_ptr().cast(),
                    v.as_mut_ptr().cast(),
                    ptr::null_mut(),
                )
            },
            ctx.raw,
        )?;
        com.truncate(ell.saturating_sub(1));
        let rows: Vec<Vec<Fr>> = v.chunks(ell.max(1)).map(<[Fr]>::to_vec).collect();
        let v: [Vec<Fr>; 3] = rows.try_into().map_err(|_| HipError::size_mismatch("three evaluation rows"))?;
        Ok(HyperKZGProof { com, w, v })
    }
}

impl CommitmentScheme for HipHyperKzg {
    type Field = Fr;
    type Proof = HyperKZGProof<Bn254>;
    type ProverSetup = HipHyperKzgSetup;
    type VerifierSetup = HyperKZGVerifierSetup<Bn254>;
    /// the committed column as it lies in HBM + the class sums begun at commit time (`crate::opening`)
    type OpeningHint = crate::opening::HipOpeningHint;
    /// The reference's parameters and the GPU to put the bases on.
    type SetupParams = (<HyperKZGScheme<Bn254> as CommitmentScheme>::SetupParams, i32);

    fn setup((params, device): Self::SetupParams) -> Result<(Self::ProverSetup, Self::VerifierSetup), OpeningsError> {
        let (inner, verifier) = <HyperKZGScheme<Bn254> as CommitmentScheme>::setup(params)?;
        let ctx = HipContext::new(device).map_err(|e| OpeningsError::CommitFailed(format!("no MI355X context: {e:?}")))?;
        let prover = HipHyperKzgSetup::new(&ctx, inner).map_err(|e| OpeningsError::CommitFailed(format!("SRS upload failed: {e:?}")))?;
        Ok((prover, verifier))
    }

    fn verifier_setup(prover_setup: &Self::ProverSetup) -> Self::VerifierSetup {
        HyperKZGVerifierSetup::from(&prover_setup.inner)
    }

    fn commit<P: MultilinearPoly<Self::Field> + ?Sized>(poly: &P, setup: &Self::ProverSetup) -> Result<(Self::Output, Self::OpeningHint), OpeningsError> {
        setup
            .with_device(|ctx, srs| Self::commit_table(ctx, srs, &Self::upload(ctx, poly)?))
            .map(|c| (c, Self::OpeningHint::default()))
            .map_err(|e| OpeningsError::CommitFailed(format!("HyperKZG commit failed on the device: {e:?}")))
    }

    fn open<P: MultilinearPoly<Self::Field> + ?Sized>(
        poly: &P,
        point: &[Self::Field],
        _eval: Self::Field,
        setup: &Self::ProverSetup,
        hint: Option<Self::OpeningHint>,
        transcript: &mut impl Transcript<Challenge = Self::Field>,
    ) -> Result<Self::Proof, OpeningsError> {
        // a homomorphic batch whose members are all resident (`combine_hints`): the joint polynomial is built and opened on the device, the host-side RLC is never densified
        if let Some(crate::opening::HipOpeningHint::Joint(joint)) = &hint {
            if joint.block.as_ref().map_or(joint.log_k, |b| b.log_k + b.log_t) == point.len() || joint.block.is_none() {
                return setup
                    .with_device(|ctx, srs| Self::open_joint(ctx, srs, joint, point, transcript))
                    .map_err(|e| OpeningsError::ProveFailed(format!("HyperKZG batch open failed on the device: {e:?}")));
            }
        }
        setup
            .with_device(|ctx, srs| Self::open_table(ctx, srs, &Self::upload(ctx, poly)?, point, transcript, &[]))
            .map_err(|e| OpeningsError::ProveFailed(format!("HyperKZG open failed on the device: {e:?}")))
    }

    fn verify(
        commitment: &Self::Output,
        point: &[Self::Field],
        eval: Self::Field,
        proof: &Self::Proof,
        setup: &Self::VerifierSetup,
        transcript: &mut impl Transcript<Challenge = Self::Field>,
    ) -> Result<(), OpeningsError> {
        <HyperKZGScheme<Bn254> as CommitmentScheme>::verify(commitment, point, eval, proof, setup, transcript)
    }
}

impl HipHyperKzg {
    /// `jolt_host_hyperkzg_open_grid`: the batch's joint polynomial (`jolt_grid_joint_polynomial` over the resident columns) opened with its first level commitments by
    /// linearity from the class sums the commit slot began (`jolt_grid_hint_begin`); the same proof `open_table` returns for the densified joint polynomial.
    fn open_joint<T: Transcript<Challenge = Fr>>(
        ctx: &Arc<HipContext>,
        srs: &HipSrs,
        joint: &crate::opening::JointHint,
        point: &[Fr],
        transcript: &mut T,
    ) -> Result<HyperKZGProof<Bn254>, HipError> {
        let table = joint.joint_polynomial(ctx)?;
        // the class sums begun at commit time, or begun now (main stream) when the batch comes without them
        let class_sums = match &joint.block {
            Some(block) => {
                let parked = block.hint.lock().unwrap_or_else(std::sync::PoisonError::into_inner).take();
                match parked {
                    Some(h) => Some(h),
                    None => Some(crate::opening::HipGridHint::begin(ctx, srs, &block.indices, 2, false)?),
                }
            }
            None => None,
        };
        let Some(class_sums) = class_sums else {
            return Self::open_table(ctx, srs, &table, point, transcript, &[]);
        };
        let ell = point.len();
        let mut com = vec![Bn254G1::default(); ell.saturating_sub(1).max(1)];
        let mut w = [Bn254G1::default(); 3];
        let mut v = vec![Fr::default(); 3 * ell.max(1)];
        let mut hook = Hook { transcript };
        let dense: Vec<*mut ffi::jolt_table> = joint.dense.iter().map(|(t, _)| t.raw).collect();
        let dense_scalars: Vec<Fr> = joint.dense.iter().map(|(_, s)| *s).collect();
        // SAFETY: live handles; output arrays sized as the header states; one scalar per hint column / dense table; the hook and its transcript outlive the call.
        check(
            unsafe {
                ffi::jolt_host_hyperkzg_open_grid(
                    ctx.raw,
                    srs.raw,
                    table.raw,
                    point.as_ptr().cast(),
                    ell,
                    0,
                    Some(open_hook::<T>),
                    (&mut hook as *mut Hook<'_, T>).cast(),
                    class_sums.raw,
                    class_sums.levels,
                    joint.onehot_scalars.as_ptr().cast(),
                    if dense.is_empty() { ptr::null() } else { dense.as_ptr() },
                    dense.len(),
                    if dense.is_empty() { ptr::null() } else { dense_scalars.as_ptr().cast() },
                    com.as_mut_ptr().cast(),
                    w.as_mut_ptr().cast(),
                    v.as_mut_ptr().cast(),
                    ptr::null_mut(),
                )
            },
            ctx.raw,
        )?;
        com.truncate(ell.saturating_sub(1));
        let rows: Vec<Vec<Fr>> = v.chunks(ell.max(1)).map(<[Fr]>::to_vec).collect();
        let v: [Vec<Fr>; 3] = rows.try_into().map_err(|_| HipError::size_mismatch("three evaluation rows"))?;
        Ok(HyperKZGProof { com, w, v })
    }
}

impl AdditivelyHomomorphic for HipHyperKzg {
    /// `schemes.rs:157-162`: the members' resident columns with the batch's RLC scalars -- what `open` builds the joint polynomial from on the device.
    fn combine_hints(hints: Vec<Self::OpeningHint>, scalars: &[Self::Field]) -> Self::OpeningHint {
        crate::opening::combine_hints(hints, scalars)
    }

    /// `C = sum_i s_i C_i` (`scheme.rs:346-352`): a few dozen points -- the host's `JoltGroup::msm`, as in the reference.
    fn combine(commitments: &[Self::Output], scalars: &[Self::Field]) -> Self::Output {
        assert_eq!(commitments.len(), scalars.len());
        let bases: Vec<Bn254G1> = commitments.iter().map(|c| c.point).collect();
        HyperKZGCommitment { point: <Bn254G1 as JoltGroup>::msm(&bases, scalars) }
    }
}
