//! The backend constructor and the bespoke (non-`PrepareKernel`) slots: `UniskipKernel` for the two uni-skip fronts
//! (`crates/jolt-kernels/src/uniskip.rs:28-54`), `CommitWitness` (`crates/jolt-kernels/src/commitment.rs:137-160`) and
//! `pub fn mi355x(ctx) -> JoltBackend<Fr, HipHyperKzg>` composing over `JoltBackend::optimized()` slot by slot, the way `optimized()` itself
//! composes over `reference()` (`crates/jolt-kernels/src/optimized/mod.rs:136-196`, registry `src/backend.rs:126-171`).
//!
//! Written blind (no Rust toolchain in the image this repository is built in) against the reference's public items.  Two of the reference's
//! helpers these slots want are crate-private today and are named where they are used: the typed-row store of `optimized/rows.rs`
//! (`BundleStore`, `SpartanOuterRow`, `SpartanProductRow`) and the constraint tables' integer forms (`extension_coefficients`,
//! `weighted_columns`).  Until `jolt-kernels` exports them (or this module moves inside it as `jolt_kernels::mi355x`), the fronts here read
//! the R1CS inputs through the PUBLIC oracle (`JoltWitnessOracle::oracle_table`) -- correct, but T x 32 bytes per input over PCIe instead of
//! the 8 / 16-byte typed columns the device kernels take (`HipInts`); the call shape on the device side is the same either way.
use std::sync::Arc;

use jolt_claims::protocols::jolt::JoltCommittedPolynomial;
use jolt_crypto::Bn254;
use jolt_field::Fr;
use jolt_hyperkzg::HyperKZGCommitment;
use jolt_kernels::commitment::{CommitWitness, CommitmentGrid, WitnessCommitment};
use jolt_kernels::uniskip::UniskipKernel;
use jolt_kernels::{JoltBackend, KernelError, ProofSession};
use jolt_poly::UnivariatePoly;
use jolt_verifier::stages::relations::{OuterRemainder, ProductRemainder};
use jolt_witness::{JoltWitnessOracle, JoltWitnessPlane};

use crate::context::{HipContext, HipTable};
use crate::member::HipPrepare;
use crate::ops::{HipHotIndices, HipInts, SpartanSums};
use crate::pcs::{HipHyperKzg, HipHyperKzgSetup};
use crate::scheduler::HipBuildRoundScheduler;

/// What a uni-skip front parks for its remainder slot (`SpartanOuterCarry`, `optimized/spartan_outer.rs:370-392`): the integer columns
/// stay in HBM between the two slots -- the remainder's Az / Bz are materialised from the SAME resident columns once the uni-skip
/// challenge is known (`SpartanSums::materialize`), nothing is uploaded twice.
pub struct HipUniskipCarry {
    pub log_t: usize,
    pub tau: Vec<Fr>,
    pub columns: Vec<HipInts>,
    pub t1_values: Vec<Fr>,
    pub streams: u32,
}

/// Integer column weights of the extended nodes: `[node][stream][1 + inputs]`, the integer Lagrange extension coefficients folded over
/// the constraint rows (outer: `extension_coefficients()` x the uniform constraint table, `optimized/spartan_outer.rs:276-370`; product:
/// `spartan_product.rs:86-105`).  Supplied by the caller that owns the constraint tables; O(rows x inputs) host work, done once.
pub struct NodeWeights {
    pub a: Vec<i64>,
    pub b: Vec<i64>,
    pub nodes: usize,
    /// position of node k inside the `2 * DOMAIN - 1` value vector the first-round polynomial is interpolated from
    pub positions: Vec<usize>,
    pub extended_size: usize,
}

/// Stage-1 / stage-2 uni-skip front on the device.  `streams` = 2: Spartan outer (cycle || stream), 1: product virtualization.
pub struct HipUniskip<R> {
    ctx: Arc<HipContext>,
    weights: NodeWeights,
    /// R1CS input ids in column order (what `columns` holds), read through the witness oracle
    inputs: Vec<jolt_witness::JoltPolynomialId>,
    streams: u32,
    /// the reference's own first-round assembly (`centered_lagrange_evals` + `interpolate_to_coeffs` + `poly_mul`,
    /// `optimized/spartan_outer.rs:496-518`), handed in because those helpers are crate-private
    assemble: fn(&[Fr], &[Fr], &[Fr]) -> Result<UnivariatePoly<Fr>, KernelError<Fr>>,
    _relation: core::marker::PhantomData<fn() -> R>,
}

impl<R> HipUniskip<R> {
    pub fn new(
        ctx: &Arc<HipContext>,
        weights: NodeWeights,
        inputs: Vec<jolt_witness::JoltPolynomialId>,
        streams: u32,
        assemble: fn(&[Fr], &[Fr], &[Fr]) -> Result<UnivariatePoly<Fr>, KernelError<Fr>>,
    ) -> Self {
        Self { ctx: Arc::clone(ctx), weights, inputs, streams, assemble, _relation: core::marker::PhantomData }
    }

    /// The R1CS inputs as device-resident integer columns.  Every input of the Jolt R1CS is a machine integer (flags, u64 registers, one
    /// i128 product): the oracle's field elements are narrowed back (`Fr::to_i128`-style canonical decode); a value that does not fit is an
    /// invariant violation of the witness, not of this backend.
    fn columns(&self, witness: &dyn JoltWitnessPlane<Fr>) -> Result<Vec<HipInts>, KernelError<Fr>> {
        self.inputs
            .iter()
            .map(|id| {
                let table = witness.oracle_table(*id).map_err(KernelError::from)?;
                let ints: Vec<i128> = table
                    .iter()
                    .map(|v| v.to_i128().ok_or(KernelError::InvariantViolation { reason: "an R1CS input outside the i128 range" }))
                    .collect::<Result<_, _>>()?;
                HipInts::from_i128(&self.ctx, &ints).map_err(KernelError::from)
            })
            .collect()
    }
}

macro_rules! impl_uniskip {
    ($relation:ty) => {
        impl UniskipKernel<Fr, $relation> for HipUniskip<$relation> {
            #[tracing::instrument(skip_all, name = "HipUniskip::prepare")]
            fn prepare(&self, session: &mut ProofSession, log_t: usize, tau: &[Fr], witness: &dyn JoltWitnessPlane<Fr>) -> Result<(), KernelError<Fr>> {
                let columns = self.columns(witness)?;
                // eq over (cycle || stream) for outer, over the cycles for product: tau_low = the first log_t + streams - 1 challenges
                let low = log_t + self.streams as usize - 1;
                if tau.len() < low {
                    return Err(KernelError::InvariantViolation { reason: "uni-skip tau shorter than the cycle domain" });
                }
                let eq: HipTable = self.ctx.eq_evals(&tau[..low], None).map_err(KernelError::from)?;
                let refs: Vec<&HipInts> = columns.iter().collect();
                let sums = SpartanSums { ctx: &self.ctx, inputs: &refs, streams: self.streams }
                    .uniskip_sums(&eq, &self.weights.a, &self.weights.b, self.weights.nodes)
                    .map_err(KernelError::from)?;
                // in-domain nodes stay zero (a satisfying witness vanishes there), as in the reference layout
                let mut t1_values = vec![Fr::default(); self.weights.extended_size];
                for (position, value) in self.weights.positions.iter().zip(sums) {
                    t1_values[*position] = value;
                }
                session.park(HipUniskipCarry { log_t, tau: tau.to_vec(), columns, t1_values, streams: self.streams });
                Ok(())
            }

            #[tracing::instrument(skip_all, name = "HipUniskip::first_round_poly")]
            fn first_round_poly(&self, session: &mut ProofSession, late_tau: &[Fr]) -> Result<UnivariatePoly<Fr>, KernelError<Fr>> {
                let carry = session
                    .state::<HipUniskipCarry>()
                    .ok_or(KernelError::InvariantViolation { reason: "the uni-skip slot parked no carry for the first-round polynomial" })?;
                (self.assemble)(&carry.tau, late_tau, &carry.t1_values)
            }
        }
    };
}
impl_uniskip!(OuterRemainder<Fr>);
impl_uniskip!(ProductRemainder<Fr>);

/// Stage 0 on the device: every trace-derived committed polynomial over the shared embedding grid
/// (`CommitmentGrid`, cycle-major placement, `crates/jolt-kernels/src/commitment.rs:86-130`): a one-hot column's commitment is the sum of
/// the T bases it selects (`jolt_grid_commit_onehot`: additions only), a dense column is one MSM of T 64-bit scalars against the SRS
/// prefix (`jolt_msm_g1_table` over the promoted column).  Modes the grid kernels do not cover (address-major order, advice) go to the
/// fallback slot, exactly as `OptimizedBackend` defers to `ReferenceBackend` (`optimized/commitment.rs:83-115`).
pub struct HipCommitWitness {
    ctx: Arc<HipContext>,
    fallback: Box<dyn CommitWitness<Fr, HipHyperKzg>>,
    /// the committed polynomial's shape: `Some(chunk_shift)` for a one-hot RA chunk of the column `source`, `None` for a dense increment column
    classify: fn(JoltCommittedPolynomial) -> Option<CommittedShape>,
}

#[derive(Clone, Copy)]
pub enum CommittedShape {
    /// RdInc / RamInc: an i64 per cycle, committed at address 0 of the grid
    Dense(jolt_witness::JoltPolynomialId),
    /// chunk `index` (log_k_chunk bits) of a lookup index / PC / RAM address column
    OneHot { source: jolt_witness::JoltPolynomialId, shift: u32 },
}

impl HipCommitWitness {
    pub fn new(ctx: &Arc<HipContext>, fallback: Box<dyn CommitWitness<Fr, HipHyperKzg>>, classify: fn(JoltCommittedPolynomial) -> Option<CommittedShape>) -> Self {
        Self { ctx: Arc::clone(ctx), fallback, classify }
    }
}

impl CommitWitness<Fr, HipHyperKzg> for HipCommitWitness {
    fn commit_witness(
        &self,
        session: &mut ProofSession,
        source: &dyn JoltWitnessPlane<Fr>,
        ids: &[JoltCommittedPolynomial],
        grid: CommitmentGrid,
        setup: &HipHyperKzgSetup,
    ) -> Result<Vec<WitnessCommitment<HipHyperKzg>>, KernelError<Fr>> {
        let shapes: Option<Vec<CommittedShape>> = ids.iter().map(|id| (self.classify)(*id)).collect();
        let Some(shapes) = shapes else {
            return self.fallback.commit_witness(session, source, ids, grid, setup);
        };
        if grid.order != jolt_kernels::commitment::TracePolynomialOrder::CycleMajor {
            return self.fallback.commit_witness(session, source, ids, grid, setup);
        }
        let cycles = 1usize << grid.log_t;
        // dense columns first onto the device (field elements, zero-extended to the grid: address 0 holds the cycles, the rest is zero) ...
        let mut dense_tables = Vec::new();
        let mut dense_slots = Vec::new();
        for (slot, shape) in shapes.iter().enumerate() {
            if let CommittedShape::Dense(col) = shape {
                let table = source.oracle_table(*col).map_err(KernelError::from)?;
                dense_tables.push(self.ctx.upload(&table[..cycles]).map_err(KernelError::from)?);
                dense_slots.push(slot);
            }
        }
        // ... their MSMs go in flight on the side lanes (three at a time) while the one-hot columns' sums of bases run on the main stream
        let mut commitments: Vec<Option<HyperKZGCommitment<Bn254>>> = vec![None; ids.len()];
        let mut onehot_done = false;
        let mut chunks = dense_tables.chunks(3).zip(dense_slots.chunks(3)).peekable();
        // (called with the setup's device lock held: the columns go straight to `HipHotIndices::grid_commit`, not through `HipHyperKzgSetup::grid_commit_onehot`)
        let mut commit_onehots = |ctx: &Arc<HipContext>, srs: &crate::msm::HipSrs, out: &mut Vec<Option<HyperKZGCommitment<Bn254>>>| -> Result<(), crate::status::HipError> {
            for (slot, shape) in shapes.iter().enumerate() {
                if let CommittedShape::OneHot { source: col, shift } = shape {
                    let table = source.oracle_table(*col).map_err(|_| crate::status::HipError::size_mismatch("one-hot source column unavailable"))?;
                    let k = 1u32 << grid.log_k_chunk;
                    let hot: Vec<u8> = table[..cycles].iter().map(|v| v.to_u64().map_or(0xFF, |a| ((a >> shift) & u64::from(k - 1)) as u8)).collect();
                    let indices = HipHotIndices::upload(ctx, &hot, 1, cycles, k)?;
                    let point = indices.grid_commit(srs)?.into_iter().next().ok_or_else(|| crate::status::HipError::size_mismatch("grid commit returned no point"))?;
                    out[slot] = Some(HyperKZGCommitment { point });
                }
            }
            Ok(())
        };
        if chunks.peek().is_none() {
            setup.with_device(|ctx, srs| commit_onehots(ctx, srs, &mut commitments)).map_err(KernelError::from)?;
            onehot_done = true;
        }
        for (tables, slots) in chunks {
            let refs: Vec<&HipTable> = tables.iter().collect();
            let first = !onehot_done;
            let mut staged = vec![None; ids.len()];
            let (coms, ()) = HipHyperKzg::commit_tables_overlapped(&refs, setup, |ctx, srs| if first { commit_onehots(ctx, srs, &mut staged) } else { Ok(()) }).map_err(KernelError::from)?;
            if first {
                for (dst, src) in commitments.iter_mut().zip(staged) {
                    if src.is_some() {
                        *dst = src;
                    }
                }
                onehot_done = true;
            }
            for (slot, com) in slots.iter().zip(coms) {
                commitments[*slot] = Some(com);
            }
        }
        ids.iter()
            .zip(commitments)
            .map(|(id, commitment)| {
                let commitment = commitment.ok_or(KernelError::InvariantViolation { reason: "a committed polynomial was left without a commitment" })?;
                Ok(WitnessCommitment { id: *id, commitment, hint: () })
            })
            .collect()
    }

    fn commit_advice(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessOracle<Fr>,
        id: JoltCommittedPolynomial,
        grid: CommitmentGrid,
        setup: &HipHyperKzgSetup,
    ) -> Result<WitnessCommitment<HipHyperKzg>, KernelError<Fr>> {
        // advice grids are small single-column commits (optimized/commitment.rs:103-114): the fallback's pass is the right shape
        self.fallback.commit_advice(session, witness, id, grid, setup)
    }
}

/// Everything the constructor cannot derive from public items of the reference (see the module docs).
pub struct Mi355xParts {
    pub outer_weights: NodeWeights,
    pub outer_inputs: Vec<jolt_witness::JoltPolynomialId>,
    pub product_weights: NodeWeights,
    pub product_inputs: Vec<jolt_witness::JoltPolynomialId>,
    pub assemble_outer: fn(&[Fr], &[Fr], &[Fr]) -> Result<UnivariatePoly<Fr>, KernelError<Fr>>,
    pub assemble_product: fn(&[Fr], &[Fr], &[Fr]) -> Result<UnivariatePoly<Fr>, KernelError<Fr>>,
    pub classify_committed: fn(JoltCommittedPolynomial) -> Option<CommittedShape>,
}

/// `JoltBackend::optimized()` with the slots this crate serves overwritten by their device kernels -- the composition pattern of
/// `optimized/mod.rs:136-196`: the commit slot, the round-traversal factory and the two uni-skip fronts here; the cycle-domain relation
/// slots of stages 2 - 6b through [`with_relation`], one line per relation with that relation's leaf resolver (`member::ResolveLeaves`:
/// which opening / derived table each leaf of its `Expr` is, `crates/jolt-kernels/src/reference/views.rs:20-138`).  A slot whose `prepare`
/// answers `KernelError::Unsupported` (no gfx950 device, a descriptor beyond the library's compiled limits, out of HBM) is recoverable: the
/// stage driver retries it against `optimized()`.
///
/// `JoltBackend::<Fr, HipHyperKzg>::optimized()` is bounded by `PCS: ModeStreamingCommitment` (`optimized/mod.rs:136-139`); `crate::streaming` is where
/// `HipHyperKzg` meets it (`StreamingCommitment`, and `ZkStreamingCommitment` for the `zk` feature).
pub fn mi355x(ctx: &Arc<HipContext>, parts: Mi355xParts) -> JoltBackend<Fr, HipHyperKzg> {
    let mut backend = JoltBackend::<Fr, HipHyperKzg>::optimized();
    let commit = std::mem::replace(&mut backend.commit, Box::new(NoCommit));
    backend.commit = Box::new(HipCommitWitness::new(ctx, commit, parts.classify_committed));
    backend.round_scheduler = Box::new(HipBuildRoundScheduler { ctx: Arc::clone(ctx) });
    backend.spartan_outer_uniskip = Box::new(HipUniskip::<OuterRemainder<Fr>>::new(ctx, parts.outer_weights, parts.outer_inputs, 2, parts.assemble_outer));
    backend.spartan_product_uniskip = Box::new(HipUniskip::<ProductRemainder<Fr>>::new(ctx, parts.product_weights, parts.product_inputs, 1, parts.assemble_product));
    backend
}

/// One relation slot onto the device: `backend.inc_claim_reduction = with_relation(ctx, LowToHigh, IncLeaves)` etc.  The generic device
/// member behind it is the twin of `NaiveSumcheckProver::new(&inputs, opening_tables, derived_tables, order)`
/// (`crates/jolt-kernels/src/reference/naive.rs:136-205`), so every cycle-domain relation of SURVEY.md section 8 a13 takes this one line.
pub fn with_relation<R, T>(ctx: &Arc<HipContext>, order: jolt_poly::BindingOrder, leaves: T) -> Box<HipPrepare<R, T>> {
    Box::new(HipPrepare::new(Arc::clone(ctx), order, leaves))
}

/// Stand-in that only ever lives for the duration of a `mem::replace`.
struct NoCommit;
impl CommitWitness<Fr, HipHyperKzg> for NoCommit {
    fn commit_witness(
        &self,
        _: &mut ProofSession,
        _: &dyn JoltWitnessPlane<Fr>,
        _: &[JoltCommittedPolynomial],
        _: CommitmentGrid,
        _: &HipHyperKzgSetup,
    ) -> Result<Vec<WitnessCommitment<HipHyperKzg>>, KernelError<Fr>> {
        Err(KernelError::Unsupported { reason: "placeholder commit slot" })
    }
    fn commit_advice(
        &self,
        _: &mut ProofSession,
        _: &dyn JoltWitnessOracle<Fr>,
        _: JoltCommittedPolynomial,
        _: CommitmentGrid,
        _: &HipHyperKzgSetup,
    ) -> Result<WitnessCommitment<HipHyperKzg>, KernelError<Fr>> {
        Err(KernelError::Unsupported { reason: "placeholder commit slot" })
    }
}
