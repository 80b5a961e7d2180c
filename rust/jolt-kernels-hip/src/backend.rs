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
use jolt_verifier::stages::stage1::outer_remainder::OuterRemainder;
use jolt_verifier::stages::stage2::product_remainder::ProductRemainder;
use jolt_witness::{JoltWitnessOracle, JoltWitnessPlane};

use crate::context::{HipContext, HipTable};
use crate::leaves;
use crate::member::HipPrepare;
use crate::opening::{HipGridHint, HipJointOpening, HipOpeningHint, ResidentCommitted, ResidentGridBlock};
use crate::ops::{HipHotIndices, HipInts, SpartanSums};
use crate::stage;
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
                    .map(|v| crate::status::fr_to_i128(v).ok_or(KernelError::InvariantViolation { reason: "an R1CS input outside the i128 range" }))
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
        if grid.order != jolt_claims::protocols::jolt::TracePolynomialOrder::CycleMajor {
            return self.fallback.commit_witness(session, source, ids, grid, setup);
        }
        let cycles = 1usize << grid.log_t;
        let k = 1u32 << grid.log_k_chunk;
        // dense columns onto the device (field elements; address 0 of the grid holds the cycles, the rest is zero) ...
        let mut dense_tables: Vec<Arc<HipTable>> = Vec::new();
        let mut dense_slots = Vec::new();
        // ... and ALL one-hot columns as ONE resident block of hot indices (n_columns x T bytes): one sum-of-bases launch commits them, the opening hint's class sums
        // run over the same block, and stage 8 builds the joint polynomial from it (crate::opening)
        let mut onehot_slots = Vec::new();
        let mut hot: Vec<u8> = Vec::new();
        for (slot, shape) in shapes.iter().enumerate() {
            match shape {
                CommittedShape::Dense(col) => {
                    let table = source.oracle_table(*col).map_err(KernelError::from)?;
                    dense_tables.push(Arc::new(self.ctx.upload(&table[..cycles]).map_err(KernelError::from)?));
                    dense_slots.push(slot);
                }
                CommittedShape::OneHot { source: col, shift } => {
                    let table = source.oracle_table(*col).map_err(KernelError::from)?;
                    hot.extend(table[..cycles].iter().map(|v| crate::status::fr_to_u64(v).map_or(0xFF, |a| ((a >> shift) & u64::from(k - 1)) as u8)));
                    onehot_slots.push(slot);
                }
            }
        }
        let mut commitments: Vec<Option<HyperKZGCommitment<Bn254>>> = vec![None; ids.len()];
        let mut hints: Vec<HipOpeningHint> = vec![HipOpeningHint::None; ids.len()];
        // the one-hot block: commit, then the class sums of the opening's first two level commitments in the BACKGROUND (jolt_grid_hint_begin: lowest-priority stream,
        // one wavefront per SIMD) -- they depend on the witness and the SRS alone and run under the latency-bound stages between here and stage 8
        let mut block: Option<Arc<ResidentGridBlock>> = None;
        let mut commit_block = |ctx: &Arc<HipContext>, srs: &crate::msm::HipSrs, out: &mut Vec<Option<HyperKZGCommitment<Bn254>>>| -> Result<(), crate::status::HipError> {
            if onehot_slots.is_empty() {
                return Ok(());
            }
            let indices = HipHotIndices::upload(ctx, &hot, onehot_slots.len(), cycles, k)?;
            for (slot, point) in onehot_slots.iter().zip(indices.grid_commit(srs)?) {
                out[*slot] = Some(HyperKZGCommitment { point });
            }
            let class_sums = HipGridHint::begin(ctx, srs, &indices, 2, true).ok(); // (a refusal costs the opening two MSMs, nothing else)
            block = Some(Arc::new(ResidentGridBlock::new(ctx, indices, onehot_slots.len(), grid.log_t, grid.log_k_chunk, class_sums)));
            Ok(())
        };
        // the dense columns' MSMs go in flight on the side lanes (three at a time) while the one-hot block's sums of bases run on the main stream
        let refs_all: Vec<&HipTable> = dense_tables.iter().map(|t| t.as_ref()).collect();
        let mut first = true;
        if refs_all.is_empty() {
            setup.with_device(|ctx, srs| commit_block(ctx, srs, &mut commitments)).map_err(KernelError::from)?;
        }
        for (tables, slots) in refs_all.chunks(3).zip(dense_slots.chunks(3)) {
            let mut staged = vec![None; ids.len()];
            let run_block = first;
            let (coms, ()) = HipHyperKzg::commit_tables_overlapped(tables, setup, |ctx, srs| if run_block { commit_block(ctx, srs, &mut staged) } else { Ok(()) }).map_err(KernelError::from)?;
            if run_block {
                for (dst, src) in commitments.iter_mut().zip(staged) {
                    if src.is_some() {
                        *dst = src;
                    }
                }
                first = false;
            }
            for (slot, com) in slots.iter().zip(coms) {
                commitments[*slot] = Some(com);
            }
        }
        if let Some(block) = &block {
            for (column, slot) in onehot_slots.iter().enumerate() {
                hints[*slot] = HipOpeningHint::OneHot { block: Arc::clone(block), column };
            }
        }
        for (table, slot) in dense_tables.iter().zip(&dense_slots) {
            hints[*slot] = HipOpeningHint::Dense { table: Arc::clone(table), log_k: grid.log_k_chunk };
        }
        // stage 8's joint-opening slot finds the resident columns here (HipJointOpening)
        let resident = session.state_or_insert_with(ResidentCommitted::default);
        for (id, hint) in ids.iter().zip(&hints) {
            let _ = resident.0.insert(*id, hint.clone());
        }
        ids.iter()
            .zip(commitments)
            .zip(hints)
            .map(|((id, commitment), hint)| {
                let commitment = commitment.ok_or(KernelError::InvariantViolation { reason: "a committed polynomial was left without a commitment" })?;
                Ok(WitnessCommitment { id: *id, commitment, hint })
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
    /// the remainder members' field weights at the uni-skip challenge, their scale and their output openings (crate-private helpers of `jolt-kernels` today)
    pub outer_remainder_weights: stage::RemainderWeights<OuterRemainder<Fr>>,
    pub outer_remainder_openings: fn(&OuterRemainder<Fr>) -> Vec<jolt_claims::protocols::jolt::JoltOpeningId>,
    pub product_remainder_weights: stage::RemainderWeights<ProductRemainder<Fr>>,
    pub product_remainder_openings: fn(&ProductRemainder<Fr>) -> Vec<jolt_claims::protocols::jolt::JoltOpeningId>,
    /// `read_raf_stage_values` of the bytecode address phase (O(K) host work from the program image; crate-private in `jolt-kernels`)
    pub bytecode_stage_values: stage::StageValues,
}

/// `JoltBackend::optimized()` with the slots this crate serves overwritten by their device kernels -- the composition pattern of
/// `optimized/mod.rs:136-196`.  27 of the registry's 37 slots (`crates/jolt-kernels/src/backend.rs:126-171`): the commit slot, the round-traversal factory, the two
/// uni-skip fronts and the joint opening; the eleven stage operators as `jolt_stage_op` kernels (`crate::stage`); the eleven cycle-domain relations of stages 2 - 6b
/// through [`with_relation`] with each relation's leaf resolver (`crate::leaves`: which opening / derived table each leaf of its `Expr` is,
/// `crates/jolt-kernels/src/reference/views.rs:20-138`).  A slot whose `prepare`
/// answers `KernelError::Unsupported` (no gfx950 device, a descriptor beyond the library's compiled limits, out of HBM) is recoverable: the
/// stage driver retries it against `optimized()`.
///
/// `JoltBackend::<Fr, HipHyperKzg>::optimized()` is bounded by `PCS: ModeStreamingCommitment` (`optimized/mod.rs:136-139`); `crate::streaming` is where
/// `HipHyperKzg` meets it (`StreamingCommitment`, and `ZkStreamingCommitment` for the `zk` feature).
pub fn mi355x(ctx: &Arc<HipContext>, parts: Mi355xParts) -> JoltBackend<Fr, HipHyperKzg> {
    #[allow(non_upper_case_globals)]
    const LowToHigh: jolt_poly::BindingOrder = jolt_poly::BindingOrder::LowToHigh;
    let mut backend = JoltBackend::<Fr, HipHyperKzg>::optimized();
    // ---- bespoke slots: commit (stage 0), the round-traversal factory, the two uni-skip fronts, the joint opening (stage 8)
    let commit = std::mem::replace(&mut backend.commit, Box::new(NoCommit));
    backend.commit = Box::new(HipCommitWitness::new(ctx, commit, parts.classify_committed));
    backend.round_scheduler = Box::new(HipBuildRoundScheduler { ctx: Arc::clone(ctx) });
    backend.spartan_outer_uniskip = Box::new(HipUniskip::<OuterRemainder<Fr>>::new(ctx, parts.outer_weights, parts.outer_inputs, 2, parts.assemble_outer));
    backend.spartan_product_uniskip = Box::new(HipUniskip::<ProductRemainder<Fr>>::new(ctx, parts.product_weights, parts.product_inputs, 1, parts.assemble_product));
    let joint = std::mem::replace(&mut backend.joint_opening, Box::new(NoJointOpening));
    backend.joint_opening = Box::new(HipJointOpening { ctx: Arc::clone(ctx), fallback: joint });
    // ---- the stage operators: one jolt_stage_op each (crate::stage)
    backend.spartan_outer_remainder =
        Box::new(stage::HipSpartanRemainder::<OuterRemainder<Fr>> { ctx: Arc::clone(ctx), weights: parts.outer_remainder_weights, openings: parts.outer_remainder_openings });
    backend.spartan_product_remainder =
        Box::new(stage::HipSpartanRemainder::<ProductRemainder<Fr>> { ctx: Arc::clone(ctx), weights: parts.product_remainder_weights, openings: parts.product_remainder_openings });
    backend.ram_read_write = Box::new(stage::HipRamReadWrite::new(ctx));
    backend.ram_raf_evaluation = Box::new(stage::HipRamRafEvaluation::new(ctx));
    backend.ram_output_check = Box::new(stage::HipRamOutputCheck::new(ctx));
    backend.registers_read_write = Box::new(stage::HipRegistersReadWrite::new(ctx));
    backend.instruction_read_raf = Box::new(stage::HipInstructionReadRaf::new(ctx));
    backend.booleanity_address = Box::new(stage::HipBooleanityAddress::new(ctx));
    backend.booleanity_cycle = Box::new(stage::HipBooleanityCycle::new(ctx));
    backend.bytecode_read_raf_address = Box::new(stage::HipBytecodeReadRafAddressWith { slot: stage::HipBytecodeReadRafAddress::new(ctx), stage_values: parts.bytecode_stage_values });
    backend.bytecode_read_raf_cycle = Box::new(stage::HipBytecodeReadRafCycle::new(ctx));
    backend.hamming_weight_claim_reduction = Box::new(stage::HipHammingWeightClaimReduction::new(ctx));
    // ---- the eleven cycle-domain relations of stages 2 - 6b: the generic device member over each relation's leaf resolver (crate::leaves)
    backend.instruction_claim_reduction = with_relation(ctx, LowToHigh, leaves::InstructionClaimReductionLeaves);
    backend.spartan_shift = with_relation(ctx, LowToHigh, leaves::SpartanShiftLeaves);
    backend.instruction_input = with_relation(ctx, LowToHigh, leaves::InstructionInputLeaves);
    backend.registers_claim_reduction = with_relation(ctx, LowToHigh, leaves::RegistersClaimReductionLeaves);
    backend.ram_val_check = with_relation(ctx, LowToHigh, leaves::RamValCheckLeaves);
    backend.registers_val_evaluation = with_relation(ctx, LowToHigh, leaves::RegistersValEvaluationLeaves);
    backend.ram_ra_claim_reduction = with_relation(ctx, LowToHigh, leaves::RamRaClaimReductionLeaves);
    backend.inc_claim_reduction = with_relation(ctx, LowToHigh, leaves::IncClaimReductionLeaves);
    backend.ram_hamming_booleanity = with_relation(ctx, LowToHigh, leaves::RamHammingBooleanityLeaves);
    backend.ram_ra_virtualization = with_relation(ctx, LowToHigh, leaves::RamRaVirtualizationLeaves);
    backend.instruction_ra_virtualization = with_relation(ctx, LowToHigh, leaves::InstructionRaVirtualizationLeaves);
    // left on `optimized()`: booleanity_cycle (its joint (address || cycle) member is harness-scale in the reference tier), the advice and
    // precommitted-program reductions and the advice opening evaluation (no T-scale work: DESIGN.md section 7)
    backend
}

/// One relation slot onto the device: `backend.inc_claim_reduction = with_relation(ctx, LowToHigh, IncLeaves)` etc.  The generic device
/// member behind it is the twin of `NaiveSumcheckProver::new(&inputs, opening_tables, derived_tables, order)`
/// (`crates/jolt-kernels/src/reference/naive.rs:136-205`), so every cycle-domain relation of SURVEY.md section 8 a13 takes this one line.
pub fn with_relation<R, T>(ctx: &Arc<HipContext>, order: jolt_poly::BindingOrder, leaves: T) -> Box<HipPrepare<R, T>> {
    Box::new(HipPrepare::new(Arc::clone(ctx), order, leaves))
}

/// Stand-in that only ever lives for the duration of a `mem::replace`.
struct NoJointOpening;
impl jolt_kernels::opening::JointOpeningPolynomials<Fr> for NoJointOpening {
    fn prepare(
        &self,
        _: &mut ProofSession,
        _: &dyn JoltWitnessPlane<Fr>,
        _: &[JoltCommittedPolynomial],
        _: &std::collections::BTreeMap<JoltCommittedPolynomial, Vec<Fr>>,
        _: CommitmentGrid,
    ) -> Result<Vec<Box<dyn jolt_poly::MultilinearPoly<Fr>>>, KernelError<Fr>> {
        Err(KernelError::Unsupported { reason: "placeholder joint-opening slot" })
    }
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
