//! Leaf resolvers of the eleven cycle-domain relations of stages 2 - 6b: which table each leaf of a relation's `Expr` is, on the device.
//!
//! The reference tier builds, per relation, `opening_tables` / `derived_tables` maps and hands them to `NaiveSumcheckProver::new` (`crates/jolt-kernels/src/reference/
//! <relation>.rs`, helpers in `reference/views.rs:20-138`).  [`crate::member::HipPrepare`] is the device twin of that constructor; a [`crate::member::ResolveLeaves`]
//! below is the device twin of one relation's two maps, leaf by leaf, each arm citing the reference lines it restates:
//!   * `dense_view(witness, id)`            -> the oracle column uploaded (`HipContext::upload`),
//!   * `eq_table(point)` / `LtPolynomial::evaluations` / `EqPlusOnePolynomial::evals(..).1` -> expanded on the device (`jolt_eq_evals`, `jolt_lt_evals`,
//!     `jolt_eq_plus_one_evals`) from the same point -- no T-sized table crosses PCIe for a derived leaf,
//!   * `address_fold(witness, id, log_t, point)` (the K x T one-hot grid folded along its address axis) -> ONE gather per cycle from the K-entry eq table of the
//!     point, over the access column the grid is the one-hot of (`optimized/lazy_ra.rs:1-33`: a selector column is a point mass per cycle); the K x T grid is never built.
//!
//! With these, `mi355x()` puts every one of the eleven slots on the device: `backend.<slot> = with_relation(ctx, LowToHigh, <Leaves>)`.
//! Written blind (no Rust toolchain in this image); the relation accessors and id constructors are the ones the cited reference files use.
use std::sync::Arc;

use jolt_claims::protocols::jolt::geometry::dimensions::{committed_address_chunks, REGISTER_ADDRESS_BITS};
use jolt_claims::protocols::jolt::{
    IncClaimReductionPublic, InstructionClaimReductionPublic, InstructionInputPublic, InstructionRaVirtualizationPublic, JoltDerivedId, JoltOpeningId, RamHammingBooleanityPublic,
    RamRaClaimReductionPublic, RamRaVirtualizationPublic, RamValCheckPublic, RegistersClaimReductionPublic, RegistersValEvaluationPublic, SpartanShiftPublic,
};
use jolt_field::Fr;
use jolt_kernels::{KernelError, ProverInputs};
use jolt_poly::EqPolynomial;
use jolt_verifier::stages::relations::ConcreteSumcheck;
use jolt_verifier::stages::stage2::instruction_claim_reduction::InstructionClaimReduction;
use jolt_verifier::stages::stage3::outputs::{InstructionInput, RegistersClaimReduction, SpartanShift};
use jolt_verifier::stages::stage4::ram_val_check::RamValCheck;
use jolt_verifier::stages::stage5::ram_ra_claim_reduction::RamRaClaimReduction;
use jolt_verifier::stages::stage5::registers_val_evaluation::RegistersValEvaluation;
use jolt_verifier::stages::stage6b::inc_claim_reduction::IncClaimReduction;
use jolt_verifier::stages::stage6b::instruction_ra_virtualization::InstructionRaVirtualization;
use jolt_verifier::stages::stage6b::ram_hamming_booleanity::RamHammingBooleanity;
use jolt_verifier::stages::stage6b::ram_ra_virtualization::RamRaVirtualization;
use jolt_witness::witnesses::{LookupIndex, RaChunkSelector, RemappedRamAddress};
use jolt_witness::{collect_bundles, JoltWitnessPlane, WitnessBundle};

use crate::context::{HipContext, HipTable};
use crate::member::ResolveLeaves;

/// `dense_view` (`reference/views.rs:20-25`): the oracle column, uploaded.
fn dense(ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
    let table = witness.oracle_table(id.polynomial_id())?;
    Ok(ctx.upload(&table)?)
}
fn eq(ctx: &Arc<HipContext>, point: &[Fr]) -> Result<HipTable, KernelError<Fr>> {
    Ok(ctx.eq_evals(point, None)?)
}
fn unknown_derived<T>() -> Result<T, KernelError<Fr>> {
    Err(KernelError::InvariantViolation { reason: "a derived leaf this relation's resolver does not know" })
}

#[derive(Clone, Copy, Debug, WitnessBundle)]
struct RamAddressBundle {
    address: RemappedRamAddress,
}
#[derive(Clone, Copy, Debug, WitnessBundle)]
struct LookupIndexBundle {
    lookup_index: LookupIndex,
}

/// `address_fold` of a one-hot grid (`reference/views.rs:31-62`) without the grid: `folded[j] = eq(point, hot(j))`, zero on a cold cycle.
fn gather_fold(ctx: &Arc<HipContext>, point: &[Fr], hot: impl Iterator<Item = Option<usize>>) -> Result<HipTable, KernelError<Fr>> {
    let weights = EqPolynomial::evals(point, None);
    let folded: Vec<Fr> = hot
        .map(|k| match k {
            Some(k) if k < weights.len() => Ok(weights[k]),
            Some(_) => Err(KernelError::InvariantViolation { reason: "a one-hot address outside the fold point's domain" }),
            None => Ok(Fr::default()),
        })
        .collect::<Result<_, _>>()?;
    Ok(ctx.upload(&folded)?)
}
fn ram_addresses(witness: &dyn JoltWitnessPlane<Fr>, log_t: usize) -> Result<Vec<Option<usize>>, KernelError<Fr>> {
    let rows: Vec<RamAddressBundle> = collect_bundles(witness, 1usize << log_t)?;
    Ok(rows.iter().map(|r| r.address.0.map(|a| a as usize)).collect())
}

/// stage 2 `instruction_claim_reduction` (`reference/instruction_claim_reduction.rs`): five reduced operand columns, `EqSpartan = eq(tau_low, .)`.
pub struct InstructionClaimReductionLeaves;
impl ResolveLeaves<InstructionClaimReduction<Fr>> for InstructionClaimReductionLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, _: &ProverInputs<'_, Fr, InstructionClaimReduction<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        dense(ctx, witness, id)
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, InstructionClaimReduction<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        if *id == JoltDerivedId::from(InstructionClaimReductionPublic::EqSpartan) {
            return eq(ctx, inputs.relation.tau_low());
        }
        unknown_derived()
    }
}

/// stage 3 `spartan_shift` (`reference/spartan_shift.rs`): five shifted columns, `EqPlusOneOuter` / `EqPlusOneProduct = EqPlusOnePolynomial::evals(point).1`.
pub struct SpartanShiftLeaves;
impl ResolveLeaves<SpartanShift<Fr>> for SpartanShiftLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, _: &ProverInputs<'_, Fr, SpartanShift<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        dense(ctx, witness, id)
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, SpartanShift<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        let point = if *id == JoltDerivedId::from(SpartanShiftPublic::EqPlusOneOuter) {
            inputs.relation.product_uniskip_tau_low()
        } else if *id == JoltDerivedId::from(SpartanShiftPublic::EqPlusOneProduct) {
            inputs.relation.product_remainder_opening_point()
        } else {
            return unknown_derived();
        };
        Ok(ctx.eq_plus_one_evals(point)?)
    }
}

/// stage 3 `instruction_input` (`reference/instruction_input.rs`): eight operand / flag columns, `EqProduct = eq(product_remainder_opening_point, .)`.
pub struct InstructionInputLeaves;
impl ResolveLeaves<InstructionInput<Fr>> for InstructionInputLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, _: &ProverInputs<'_, Fr, InstructionInput<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        dense(ctx, witness, id)
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, InstructionInput<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        if *id == JoltDerivedId::from(InstructionInputPublic::EqProduct) {
            return eq(ctx, inputs.relation.product_remainder_opening_point());
        }
        unknown_derived()
    }
}

/// stage 3 `registers_claim_reduction` (`reference/registers_claim_reduction.rs`): three reduced value columns, `EqSpartan = eq(product_uniskip_tau_low, .)`.
pub struct RegistersClaimReductionLeaves;
impl ResolveLeaves<RegistersClaimReduction<Fr>> for RegistersClaimReductionLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, _: &ProverInputs<'_, Fr, RegistersClaimReduction<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        dense(ctx, witness, id)
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, RegistersClaimReduction<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        if *id == JoltDerivedId::from(RegistersClaimReductionPublic::EqSpartan) {
            return eq(ctx, inputs.relation.product_uniskip_tau_low());
        }
        unknown_derived()
    }
}

/// stage 4 `ram_val_check` (`reference/ram_val_check.rs`): `ram_inc` dense, `ram_ra` = the RAM grid folded at the address half of the consumed `ram_val` point,
/// `LtCyclePlusGamma = LT(., r_cycle) + gamma`.
pub struct RamValCheckLeaves;
impl ResolveLeaves<RamValCheck<Fr>> for RamValCheckLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, inputs: &ProverInputs<'_, Fr, RamValCheck<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        use jolt_claims::protocols::jolt::geometry::ram::ram_ra_val_check;
        if *id != ram_ra_val_check() {
            return dense(ctx, witness, id);
        }
        let log_t = inputs.relation.trace_dimensions().log_t();
        let (r_address, _) = inputs.points.ram_val.split_at(inputs.relation.ram_log_k());
        gather_fold(ctx, r_address, ram_addresses(witness, log_t)?.into_iter())
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, RamValCheck<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        if *id == JoltDerivedId::from(RamValCheckPublic::LtCyclePlusGamma) {
            let (_, r_cycle) = inputs.points.ram_val.split_at(inputs.relation.ram_log_k());
            return Ok(ctx.lt_evals_plus(r_cycle, inputs.challenges.gamma)?);
        }
        unknown_derived()
    }
}

/// stage 5 `registers_val_evaluation` (`reference/registers_val_evaluation.rs`): `rd_inc` dense, `rd_wa` = the write-address grid folded at the register half of the
/// consumed `registers_val` point, `LtCycle = LT(., r_cycle)`.
pub struct RegistersValEvaluationLeaves;
impl ResolveLeaves<RegistersValEvaluation<Fr>> for RegistersValEvaluationLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, inputs: &ProverInputs<'_, Fr, RegistersValEvaluation<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        use jolt_claims::protocols::jolt::geometry::registers::rd_wa_val_evaluation;
        if *id != rd_wa_val_evaluation() {
            return dense(ctx, witness, id);
        }
        // the grid is REGISTER_ADDRESS_BITS x T (address-major): fold it on the host side of the oracle as the reference does, 128 x T entries, then upload the T results
        let log_t = inputs.relation.trace_dimensions().log_t();
        let (r_address, _) = inputs.points.registers_val.split_at(REGISTER_ADDRESS_BITS);
        let grid = witness.oracle_table(id.polynomial_id())?;
        let weights = EqPolynomial::evals(r_address, None);
        let cycles = 1usize << log_t;
        if grid.len() != weights.len() << log_t {
            return Err(KernelError::TableSizeMismatch { table: format!("{id:?}"), expected: weights.len() << log_t, got: grid.len() });
        }
        let folded: Vec<Fr> = (0..cycles).map(|j| (0..weights.len()).map(|k| grid[(k << log_t) | j] * weights[k]).sum()).collect();
        Ok(ctx.upload(&folded)?)
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, RegistersValEvaluation<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        if *id == JoltDerivedId::from(RegistersValEvaluationPublic::LtCycle) {
            let (_, r_cycle) = inputs.points.registers_val.split_at(REGISTER_ADDRESS_BITS);
            return Ok(ctx.lt_evals(r_cycle)?);
        }
        unknown_derived()
    }
}

/// stage 5 `ram_ra_claim_reduction` (`reference/ram_ra_claim_reduction.rs`): `ram_ra` folded at the relation's reduced address point, three eq tables at the cycle
/// halves of the consumed RAF / read-write / value-check points.
pub struct RamRaClaimReductionLeaves;
impl ResolveLeaves<RamRaClaimReduction<Fr>> for RamRaClaimReductionLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, inputs: &ProverInputs<'_, Fr, RamRaClaimReduction<Fr>>, _id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        let log_t = inputs.relation.trace_dimensions().log_t();
        let ram_log_k = inputs.relation.ram_log_k();
        // the three consumed points share their address half (ram_ra_claim_reduction.rs:12-31 checks it); the fold uses it
        gather_fold(ctx, &inputs.points.raf()[..ram_log_k], ram_addresses(witness, log_t)?.into_iter())
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, RamRaClaimReduction<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        let ram_log_k = inputs.relation.ram_log_k();
        let point = if *id == JoltDerivedId::from(RamRaClaimReductionPublic::EqCycleRaf) {
            &inputs.points.raf()[ram_log_k..]
        } else if *id == JoltDerivedId::from(RamRaClaimReductionPublic::EqCycleReadWrite) {
            &inputs.points.read_write()[ram_log_k..]
        } else if *id == JoltDerivedId::from(RamRaClaimReductionPublic::EqCycleValCheck) {
            &inputs.points.val_check()[ram_log_k..]
        } else {
            return unknown_derived();
        };
        eq(ctx, point)
    }
}

/// stage 6b `inc_claim_reduction` (`reference/inc_claim_reduction.rs:29-79`): RamInc / RdInc dense, four eq tables at the relation's cycle points in `publics` order.
pub struct IncClaimReductionLeaves;
impl ResolveLeaves<IncClaimReduction<Fr>> for IncClaimReductionLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, _: &ProverInputs<'_, Fr, IncClaimReduction<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        dense(ctx, witness, id)
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, IncClaimReduction<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        let publics = [
            IncClaimReductionPublic::EqRamReadWrite,
            IncClaimReductionPublic::EqRamValCheck,
            IncClaimReductionPublic::EqRegistersReadWrite,
            IncClaimReductionPublic::EqRegistersValEvaluation,
        ];
        let cycle_points = inputs.relation.cycle_points();
        for (public, point) in publics.into_iter().zip(cycle_points) {
            if *id == JoltDerivedId::from(public) {
                if point.len() != inputs.relation.rounds() {
                    return Err(KernelError::InvariantViolation { reason: "increment reduction cycle point has the wrong variable count" });
                }
                return eq(ctx, point);
            }
        }
        unknown_derived()
    }
}

/// stage 6b `ram_hamming_booleanity` (`reference/ram_hamming_booleanity.rs`): the Hamming-weight column dense, `EqCycle = eq(stage-1 cycle binding, .)`.
pub struct RamHammingBooleanityLeaves;
impl ResolveLeaves<RamHammingBooleanity<Fr>> for RamHammingBooleanityLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, _: &ProverInputs<'_, Fr, RamHammingBooleanity<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        dense(ctx, witness, id)
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, RamHammingBooleanity<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        if *id == JoltDerivedId::from(RamHammingBooleanityPublic::EqCycle) {
            return eq(ctx, inputs.relation.stage1_cycle_binding());
        }
        unknown_derived()
    }
}

/// stage 6b `ram_ra_virtualization` (`reference/ram_ra_virtualization.rs`): committed chunk i of the RAM address folded at chunk i of the reduced address point
/// (`committed_address_chunks`), `EqCycle = eq(ram_reduced_cycle, .)`.
pub struct RamRaVirtualizationLeaves;
impl ResolveLeaves<RamRaVirtualization<Fr>> for RamRaVirtualizationLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, inputs: &ProverInputs<'_, Fr, RamRaVirtualization<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        use jolt_claims::protocols::jolt::geometry::ram::committed_ram_ra;
        let relation = inputs.relation;
        let chunk_bits = relation.committed_chunk_bits();
        let chunks = committed_address_chunks(relation.ram_reduced_address(), chunk_bits);
        let index = (0..chunks.len()).find(|i| *id == committed_ram_ra(*i)).ok_or(KernelError::InvariantViolation { reason: "an opening leaf that is not a committed RAM RA chunk" })?;
        let selector = RaChunkSelector::new(index, chunks.len(), chunk_bits)?;
        let addresses = ram_addresses(witness, relation.ram_reduced_cycle().len())?;
        gather_fold(ctx, &chunks[index], addresses.into_iter().map(|a| a.map(|a| selector.chunk_u128(a as u128))))
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, RamRaVirtualization<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        if *id == JoltDerivedId::from(RamRaVirtualizationPublic::EqCycle) {
            return eq(ctx, inputs.relation.ram_reduced_cycle());
        }
        unknown_derived()
    }
}

/// stage 6b `instruction_ra_virtualization` (`reference/instruction_ra_virtualization.rs`): committed chunk i of the lookup index folded at chunk i of the instruction
/// address point, `EqCycle = eq(instruction_read_raf_cycle, .)`.
pub struct InstructionRaVirtualizationLeaves;
impl ResolveLeaves<InstructionRaVirtualization<Fr>> for InstructionRaVirtualizationLeaves {
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, inputs: &ProverInputs<'_, Fr, InstructionRaVirtualization<Fr>>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>> {
        use jolt_claims::protocols::jolt::geometry::instruction::committed_instruction_ra;
        let relation = inputs.relation;
        let chunk_bits = relation.committed_chunk_bits();
        let chunks = committed_address_chunks(relation.instruction_address(), chunk_bits);
        let index = (0..chunks.len()).find(|i| *id == committed_instruction_ra(*i)).ok_or(KernelError::InvariantViolation { reason: "an opening leaf that is not a committed instruction RA chunk" })?;
        let selector = RaChunkSelector::new(index, chunks.len(), chunk_bits)?;
        let rows: Vec<LookupIndexBundle> = collect_bundles(witness, 1usize << relation.instruction_read_raf_cycle().len())?;
        gather_fold(ctx, &chunks[index], rows.iter().map(|r| Some(selector.chunk_u128(r.lookup_index.0))))
    }
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, InstructionRaVirtualization<Fr>>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>> {
        if *id == JoltDerivedId::from(InstructionRaVirtualizationPublic::EqCycle) {
            return eq(ctx, inputs.relation.instruction_read_raf_cycle());
        }
        unknown_derived()
    }
}
