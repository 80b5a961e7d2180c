//! The stage operators behind their `JoltBackend` slots (`crates/jolt-kernels/src/backend.rs:126-171`): one `PrepareKernel` type per slot over a
//! `jolt_stage_op` of `libjolt_hip.so` (`include/jolt_hip.h`, "Stage operators as ProveRounds objects"; `jolt_amd/csrc/stage_ops.hip`).
//!
//! The C object IS the kernel: its constructor runs everything `prepare` runs in the reference's optimized tier (the T-scale passes that depend on no round
//! challenge), `prove_round` / `finish_rounds` follow the fused `ProveRounds` contract (`crates/jolt-sumcheck/src/prover.rs:45-72`) and return the round message as
//! `UnivariatePoly` coefficients, `output_claims` are `SumcheckKernel::output_claims` (`crates/jolt-kernels/src/kernel.rs:86-92`).  What this file adds is the
//! reference-side typing: which relation accessor feeds which constructor argument (each `prepare` cites the optimized slot it replaces) and which output value is
//! which field of the relation's output-claims struct.
//!
//! The witness reaches the device ONCE per proof as typed columns ([`ResidentTrace`], parked in the `ProofSession` like the reference's `RamAccessColumns::shared` /
//! `PcRow::shared` / `InstructionCycleRow::shared`, `optimized/ram_trace.rs:87-120`): bundles collected through the public `jolt_witness::collect_bundles`, uploaded as
//! `HipInts` / `HipHotIndices` / `HipReadRaf`, and shared by every slot of every stage.
//!
//! Written blind (no Rust toolchain in the image this repository is built in); `tools/rust_seam_audit.py` checks trait items, arity, bounds and the slot / relation
//! pairing against the reference's `backend.rs`.
use std::ptr;
use std::sync::Arc;

use jolt_claims::protocols::jolt::geometry::dimensions::{committed_address_chunks, REGISTER_ADDRESS_BITS};
use jolt_claims::protocols::jolt::relations::booleanity::address_phase::BooleanityAddressPhaseOutputClaims;
use jolt_claims::protocols::jolt::relations::bytecode::read_raf_address_phase::BytecodeReadRafAddressPhaseOutputClaims;
use jolt_claims::protocols::jolt::relations::instruction::read_raf::InstructionReadRafOutputClaims;
use jolt_claims::protocols::jolt::relations::ram::output_check::RamOutputCheckOutputClaims;
use jolt_claims::protocols::jolt::relations::ram::raf_evaluation::RamRafEvaluationOutputClaims;
use jolt_claims::protocols::jolt::relations::ram::read_write_checking::RamReadWriteOutputClaims;
use jolt_claims::protocols::jolt::relations::registers::read_write_checking::RegistersReadWriteOutputClaims;
use jolt_claims::protocols::jolt::{JoltChallengeId, JoltOpeningId, JoltPolynomialId, JoltRelationId, JoltVirtualPolynomial};
use jolt_claims::{InputClaims, OutputClaims, SumcheckChallenges};
use jolt_field::Fr;
use jolt_kernels::{KernelError, PrepareKernel, ProofSession, ProverInputs, SumcheckKernel, SumcheckKernelError};
use jolt_poly::UnivariatePoly;
use jolt_sumcheck::{ProveRounds, SumcheckError};
use jolt_verifier::stages::relations::{ConcreteSumcheck, ConcreteSumcheckChallenges, SumcheckInputClaims, SumcheckOutputClaims};
use jolt_verifier::stages::stage1::outer_remainder::OuterRemainder;
use jolt_verifier::stages::stage2::product_remainder::ProductRemainder;
use jolt_verifier::stages::stage2::ram_output_check::RamOutputCheck;
use jolt_verifier::stages::stage2::ram_raf_evaluation::RamRafEvaluation;
use jolt_verifier::stages::stage2::ram_read_write_checking::RamReadWriteChecking;
use jolt_verifier::stages::stage4::registers_read_write_checking::RegistersReadWriteChecking;
use jolt_verifier::stages::stage5::InstructionReadRaf;
use jolt_verifier::stages::stage6a::booleanity::BooleanityAddressPhase;
use jolt_verifier::stages::stage6a::bytecode_read_raf::BytecodeReadRafAddressPhase;
use jolt_verifier::stages::stage6b::booleanity::Booleanity;
use jolt_verifier::stages::stage6b::bytecode_read_raf::BytecodeReadRafCycle;
use jolt_verifier::stages::stage7::hamming_weight_claim_reduction::HammingWeightClaimReduction;
use jolt_witness::witnesses::{
    BytecodePc, InstructionRafFlag, LookupIndex, MappedPc, RaChunkSelector, RamInc, RamReadValue, RamWriteValue, RemappedRamAddress, TableIndex, WitnessEnv,
};
use jolt_witness::__private::TraceRow;
use jolt_witness::{collect_bundles, JoltWitnessPlane, WitnessBundle, WitnessError};

use crate::backend::HipUniskipCarry;
use crate::context::HipContext;
use crate::ffi;
use crate::ops::{HipHotIndices, HipInts, HipKeyIndex, HipReadRaf};
use crate::status::{check, to_kernel_seam_error, to_sumcheck_error, HipError};

// ------------------------------------------------------------------------------------------------------------------
// the object
// ------------------------------------------------------------------------------------------------------------------
/// An owned `jolt_stage_op`.
pub struct HipStageOp {
    ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_stage_op,
    rounds: usize,
    degree: usize,
    bound: usize,
}
// SAFETY: see HipContext: one thread at a time (`prove_batch` is single-threaded), the handle owns its device and host state.
unsafe impl Send for HipStageOp {}

impl HipStageOp {
    /// Wrap a handle a `jolt_stage_*_create` call returned.
    pub(crate) fn adopt(ctx: &Arc<HipContext>, raw: *mut ffi::jolt_stage_op) -> Result<Self, HipError> {
        let (mut rounds, mut degree) = (0usize, 0usize);
        // SAFETY: live handle, valid out-pointers.
        check(unsafe { ffi::jolt_stage_op_num_rounds(raw, &mut rounds) }, ctx.raw)?;
        // SAFETY: as above.
        check(unsafe { ffi::jolt_stage_op_degree(raw, &mut degree) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw, rounds, degree, bound: 0 })
    }

    pub fn output_values(&mut self) -> Result<Vec<Fr>, HipError> {
        let mut out = vec![Fr::default(); 256];
        let mut n = 0usize;
        // SAFETY: `out` holds 256 elements of jolt_fr_t layout; the library reports how many it wrote.
        check(unsafe { ffi::jolt_stage_op_output_claims(self.raw, out.as_mut_ptr().cast(), out.len(), &mut n) }, self.ctx.raw)?;
        out.truncate(n);
        Ok(out)
    }
}

impl ProveRounds<Fr> for HipStageOp {
    fn num_rounds(&self) -> usize {
        self.rounds
    }

    fn prove_round(&mut self, bind: Option<Fr>, round: usize, previous_claim: Fr) -> Result<UnivariatePoly<Fr>, SumcheckError<Fr>> {
        let mut coefficients = vec![Fr::default(); self.degree + 1];
        let mut n = 0usize;
        let bind_ptr = bind.as_ref().map_or(ptr::null(), |b| (b as *const Fr).cast());
        // SAFETY: live handle; `coefficients` holds degree + 1 elements of jolt_fr_t layout; bind is NULL or one element; the claim is one element.
        let status = unsafe {
            ffi::jolt_stage_op_prove_round(self.raw, bind_ptr, round, (&previous_claim as *const Fr).cast(), coefficients.as_mut_ptr().cast(), coefficients.len(), &mut n)
        };
        if status == ffi::JOLT_ERR_ROUND_CHECK {
            // the operator's own message failed s(0) + s(1) = claim (the reference kernels' hard self-check, naive.rs:298-306)
            return Err(SumcheckError::RoundCheckFailed { round, expected: previous_claim, actual: Fr::default() });
        }
        check(status, self.ctx.raw).map_err(to_sumcheck_error)?;
        if bind.is_some() {
            self.bound += 1;
        }
        coefficients.truncate(n);
        Ok(UnivariatePoly::new(coefficients))
    }

    fn finish_rounds(&mut self, bind: Fr) -> Result<(), SumcheckError<Fr>> {
        // SAFETY: live handle; one element of jolt_fr_t layout.
        check(unsafe { ffi::jolt_stage_op_finish_rounds(self.raw, (&bind as *const Fr).cast()) }, self.ctx.raw).map_err(to_sumcheck_error)?;
        self.bound += 1;
        Ok(())
    }
}

impl Drop for HipStageOp {
    fn drop(&mut self) {
        // SAFETY: owned handle, destroyed once.
        let _ = unsafe { ffi::jolt_stage_op_destroy(self.raw) };
    }
}

/// How a slot turns the operator's output values (in the order its constructor documents) into the relation's typed output claims.
type Extract<R> = Box<dyn Fn(&[Fr], &SumcheckInputClaims<Fr, R>) -> Result<SumcheckOutputClaims<Fr, R>, SumcheckKernelError<Fr>> + Send>;

/// The typed kernel of a stage-operator slot: a [`HipStageOp`] plus the slot's claim extraction.  `keep`: device inputs the operator borrows.
pub struct HipStageKernel<R: ConcreteSumcheck<Fr>>
where
    SumcheckInputClaims<Fr, R>: InputClaims<Fr>,
    SumcheckOutputClaims<Fr, R>: OutputClaims<Fr>,
    ConcreteSumcheckChallenges<Fr, R>: SumcheckChallenges<Fr, JoltChallengeId>,
{
    op: HipStageOp,
    extract: Extract<R>,
    /// what `park_residue` moves into the session (the bytecode address phase parks its operator for the cycle phase)
    park: Option<fn(HipStageOp, &mut ProofSession)>,
    _keep: Vec<Arc<dyn core::any::Any + Send + Sync>>,
}

#[cfg(feature = "allocative")]
impl<R: ConcreteSumcheck<Fr>> allocative::Allocative for HipStageKernel<R>
where
    SumcheckInputClaims<Fr, R>: InputClaims<Fr>,
    SumcheckOutputClaims<Fr, R>: OutputClaims<Fr>,
    ConcreteSumcheckChallenges<Fr, R>: SumcheckChallenges<Fr, JoltChallengeId>,
{
    fn visit<'a, 'b: 'a>(&self, visitor: &'a mut allocative::Visitor<'b>) {
        let mut visitor = visitor.enter_self_sized::<Self>();
        visitor.visit_simple(allocative::Key::new("heap"), 0usize); // the operator's state lives in HBM and inside libjolt_hip.so
        visitor.exit();
    }
}

impl<R: ConcreteSumcheck<Fr>> ProveRounds<Fr> for HipStageKernel<R>
where
    SumcheckInputClaims<Fr, R>: InputClaims<Fr>,
    SumcheckOutputClaims<Fr, R>: OutputClaims<Fr>,
    ConcreteSumcheckChallenges<Fr, R>: SumcheckChallenges<Fr, JoltChallengeId>,
{
    fn num_rounds(&self) -> usize {
        self.op.num_rounds()
    }
    fn prove_round(&mut self, bind: Option<Fr>, round: usize, previous_claim: Fr) -> Result<UnivariatePoly<Fr>, SumcheckError<Fr>> {
        self.op.prove_round(bind, round, previous_claim)
    }
    fn finish_rounds(&mut self, bind: Fr) -> Result<(), SumcheckError<Fr>> {
        self.op.finish_rounds(bind)
    }
}

impl<R: ConcreteSumcheck<Fr>> SumcheckKernel<Fr> for HipStageKernel<R>
where
    SumcheckInputClaims<Fr, R>: InputClaims<Fr>,
    SumcheckOutputClaims<Fr, R>: OutputClaims<Fr>,
    ConcreteSumcheckChallenges<Fr, R>: SumcheckChallenges<Fr, JoltChallengeId>,
{
    type Relation = R;

    fn output_claims(&mut self, inputs: &SumcheckInputClaims<Fr, R>) -> Result<SumcheckOutputClaims<Fr, R>, SumcheckKernelError<Fr>> {
        let remaining = self.op.rounds - self.op.bound.min(self.op.rounds);
        let values = self.op.output_values().map_err(|e| to_kernel_seam_error(e, remaining))?;
        (self.extract)(&values, inputs)
    }

    fn park_residue(self: Box<Self>, session: &mut ProofSession) {
        let HipStageKernel { op, park, .. } = *self;
        if let Some(park) = park {
            park(op, session);
        }
    }
}

fn kernel<R>(op: HipStageOp, keep: Vec<Arc<dyn core::any::Any + Send + Sync>>, extract: Extract<R>) -> Box<dyn SumcheckKernel<Fr, Relation = R>>
where
    R: ConcreteSumcheck<Fr> + 'static,
    SumcheckInputClaims<Fr, R>: InputClaims<Fr>,
    SumcheckOutputClaims<Fr, R>: OutputClaims<Fr>,
    ConcreteSumcheckChallenges<Fr, R>: SumcheckChallenges<Fr, JoltChallengeId>,
{
    Box::new(HipStageKernel { op, extract, park: None, _keep: keep })
}

fn short(values: &[Fr], need: usize) -> Result<(), SumcheckKernelError<Fr>> {
    if values.len() < need {
        return Err(SumcheckKernelError::InvariantViolation { reason: "a stage operator returned fewer output values than its slot reads" });
    }
    Ok(())
}

// ------------------------------------------------------------------------------------------------------------------
// the trace, resident in HBM once per proof
// ------------------------------------------------------------------------------------------------------------------
#[derive(Clone, Copy, Debug, WitnessBundle)]
struct RamAccessBundle {
    address: RemappedRamAddress,
    pre_value: RamReadValue,
    post_value: RamWriteValue,
    inc: RamInc,
}

#[derive(Clone, Copy, Debug, WitnessBundle)]
struct LookupBundle {
    lookup_index: LookupIndex,
    table: TableIndex,
    raf: InstructionRafFlag,
}

#[derive(Clone, Copy, Debug, WitnessBundle)]
struct PcBundle {
    bytecode_pc: BytecodePc,
    mapped_pc: MappedPc,
}

/// Per-cycle register activity: hand-implemented like the reference's `RegisterCycleRow` (`optimized/registers_read_write/rows.rs:22-78`) because no witness newtype
/// exposes the operand INDICES.
#[derive(Clone, Copy, Debug, Default)]
struct RegisterBundle {
    rs1: Option<(u8, u64)>,
    rs2: Option<(u8, u64)>,
    rd: Option<(u8, u64, u64)>,
    rd_inc: i128,
}

impl WitnessBundle for RegisterBundle {
    fn from_row(row: &TraceRow, _next: Option<&TraceRow>, _env: &WitnessEnv<'_>) -> Result<Self, WitnessError> {
        let rd = row.registers.rd.map(|w| (w.register, w.pre_value, w.post_value));
        Ok(Self {
            rs1: row.registers.rs1.map(|r| (r.register, r.value)),
            rs2: row.registers.rs2.map(|r| (r.register, r.value)),
            rd,
            rd_inc: rd.map_or(0, |(_, pre, post)| i128::from(post) - i128::from(pre)),
        })
    }
    fn annotated_ids() -> Vec<JoltPolynomialId> {
        Vec::new()
    }
}

/// RAM access columns on the device (`RamAccessColumns`, `optimized/ram_trace.rs:22-75`) plus RamInc and the initial memory.
pub struct ResidentRam {
    pub addresses: HipInts,
    pub pre_values: HipInts,
    pub post_values: HipInts,
    pub inc: HipInts,
    pub val_init: HipInts,
    pub val_init_host: Vec<u64>,
}
/// Register rows on the device (`RegisterCycleRow`): the hot-index columns rs1, rs2, rd and the value columns.
pub struct ResidentRegisters {
    pub indices: HipHotIndices,
    pub rs1_val: HipInts,
    pub rs2_val: HipInts,
    pub rd_pre: HipInts,
    pub rd_post: HipInts,
    pub rd_inc: HipInts,
}
/// Lookup rows on the device (`InstructionCycleRow`, `optimized/instruction_read_raf.rs:86`) and the packed flag-claim columns.
pub struct ResidentLookups {
    pub rows: HipReadRaf,
    pub claim_columns: HipHotIndices,
}
/// The PC column and its committed chunks (`PcRow`, `optimized/bytecode_read_raf.rs:80-150`).
pub struct ResidentPc {
    pub push_pc: HipInts,
    pub chunks: HipHotIndices,
    pub first_pc: u64,
    pub chunk_bits: u32,
}

// SAFETY (all four): device handles owned by the value, used from one thread at a time (see HipContext).
unsafe impl Sync for ResidentRam {}
unsafe impl Sync for ResidentRegisters {}
unsafe impl Sync for ResidentLookups {}
unsafe impl Sync for ResidentPc {}
unsafe impl Send for ResidentLookups {}

/// The session entry: every family is collected and uploaded on first request and shared afterwards.
#[derive(Default)]
pub struct ResidentTrace {
    ram: Option<Arc<ResidentRam>>,
    registers: Option<Arc<ResidentRegisters>>,
    lookups: Option<Arc<ResidentLookups>>,
    pc: Option<Arc<ResidentPc>>,
    ra_columns: Option<Arc<HipHotIndices>>,
}
crate::status::zero_host_heap!(ResidentTrace);

const NO_ACCESS: u64 = u64::MAX;

fn witness_error(e: WitnessError) -> KernelError<Fr> {
    KernelError::from(e)
}

impl ResidentTrace {
    fn entry(session: &mut ProofSession) -> &mut Self {
        session.state_or_insert_with(Self::default)
    }

    /// `RamAccessColumns::shared` (`ram_trace.rs:87-120`) + `reconstruct_val_init` (the word an address holds before its first access, `val_final` where it has none).
    pub fn ram(session: &mut ProofSession, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, log_t: usize, log_k: usize) -> Result<Arc<ResidentRam>, KernelError<Fr>> {
        if let Some(ram) = &Self::entry(session).ram {
            return Ok(Arc::clone(ram));
        }
        let cycles = 1usize << log_t;
        let bundles: Vec<RamAccessBundle> = collect_bundles(witness, cycles).map_err(witness_error)?;
        let val_final = witness.oracle_table(JoltPolynomialId::Virtual(JoltVirtualPolynomial::RamValFinal)).map_err(witness_error)?;
        if val_final.len() != 1usize << log_k {
            return Err(KernelError::TableSizeMismatch { table: "RamValFinal".to_owned(), expected: 1usize << log_k, got: val_final.len() });
        }
        let mut val_init: Vec<u64> =
            val_final.iter().map(|v| crate::status::fr_to_u64(v).ok_or(KernelError::InvariantViolation { reason: "a RAM word outside the u64 range" })).collect::<Result<_, _>>()?;
        let mut seen = vec![false; val_init.len()];
        let (mut addresses, mut pre, mut post, mut inc) = (Vec::with_capacity(cycles), Vec::with_capacity(cycles), Vec::with_capacity(cycles), Vec::with_capacity(cycles));
        for b in &bundles {
            let a = b.address.0.unwrap_or(NO_ACCESS);
            if a != NO_ACCESS {
                let k = usize::try_from(a).ok().filter(|k| *k < val_init.len()).ok_or(KernelError::InvariantViolation { reason: "a RAM address outside the padded domain" })?;
                if !seen[k] {
                    seen[k] = true;
                    val_init[k] = b.pre_value.0;
                }
            }
            addresses.push(a);
            pre.push(b.pre_value.0);
            post.push(b.post_value.0);
            inc.push(i64::try_from(b.inc.0).map_err(|_| KernelError::InvariantViolation { reason: "RamInc outside the i64 range" })?);
        }
        let ram = Arc::new(ResidentRam {
            addresses: HipInts::from_u64(ctx, &addresses)?,
            pre_values: HipInts::from_u64(ctx, &pre)?,
            post_values: HipInts::from_u64(ctx, &post)?,
            inc: HipInts::from_i64(ctx, &inc)?,
            val_init: HipInts::from_u64(ctx, &val_init)?,
            val_init_host: val_init,
        });
        Self::entry(session).ram = Some(Arc::clone(&ram));
        Ok(ram)
    }

    /// `CollectRegisterEntries::collect` (`registers_read_write/rows.rs`), as columns.
    pub fn registers(session: &mut ProofSession, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, log_t: usize) -> Result<Arc<ResidentRegisters>, KernelError<Fr>> {
        if let Some(r) = &Self::entry(session).registers {
            return Ok(Arc::clone(r));
        }
        let cycles = 1usize << log_t;
        let bundles: Vec<RegisterBundle> = collect_bundles(witness, cycles).map_err(witness_error)?;
        let mut hot = vec![0xFFu8; 3 * cycles];
        let (mut rs1_val, mut rs2_val, mut rd_pre, mut rd_post, mut rd_inc) = (vec![0u64; cycles], vec![0u64; cycles], vec![0u64; cycles], vec![0u64; cycles], vec![0i128; cycles]);
        for (j, b) in bundles.iter().enumerate() {
            if let Some((r, v)) = b.rs1 {
                hot[j] = r;
                rs1_val[j] = v;
            }
            if let Some((r, v)) = b.rs2 {
                hot[cycles + j] = r;
                rs2_val[j] = v;
            }
            if let Some((r, pre, post)) = b.rd {
                hot[2 * cycles + j] = r;
                rd_pre[j] = pre;
                rd_post[j] = post;
            }
            rd_inc[j] = b.rd_inc;
        }
        let r = Arc::new(ResidentRegisters {
            indices: HipHotIndices::upload(ctx, &hot, 3, cycles, 1u32 << REGISTER_ADDRESS_BITS)?,
            rs1_val: HipInts::from_u64(ctx, &rs1_val)?,
            rs2_val: HipInts::from_u64(ctx, &rs2_val)?,
            rd_pre: HipInts::from_u64(ctx, &rd_pre)?,
            rd_post: HipInts::from_u64(ctx, &rd_post)?,
            rd_inc: HipInts::from_i128(ctx, &rd_inc)?,
        });
        Self::entry(session).registers = Some(Arc::clone(&r));
        Ok(r)
    }

    /// `InstructionCycleRow::shared`: the lookup rows, and the output-claim facts packed as four K = 16 hot-index columns (tables 0..15 / 16..31 / 32..41, RAF rows on 0).
    pub fn lookups(session: &mut ProofSession, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, log_t: usize) -> Result<Arc<ResidentLookups>, KernelError<Fr>> {
        if let Some(l) = &Self::entry(session).lookups {
            return Ok(Arc::clone(l));
        }
        let cycles = 1usize << log_t;
        let bundles: Vec<LookupBundle> = collect_bundles(witness, cycles).map_err(witness_error)?;
        let index: Vec<u128> = bundles.iter().map(|b| b.lookup_index.0).collect();
        let table: Vec<u8> = bundles.iter().map(|b| b.table.0.map_or(0xFF, |t| t as u8)).collect();
        let raf: Vec<bool> = bundles.iter().map(|b| b.raf.0).collect();
        let mut claim = vec![0xFFu8; 4 * cycles];
        for j in 0..cycles {
            if table[j] != 0xFF {
                claim[usize::from(table[j] / 16) * cycles + j] = table[j] % 16;
            }
            if raf[j] {
                claim[3 * cycles + j] = 0;
            }
        }
        let l = Arc::new(ResidentLookups { rows: HipReadRaf::new(ctx, &index, &table, &raf, 42)?, claim_columns: HipHotIndices::upload(ctx, &claim, 4, cycles, 16)? });
        Self::entry(session).lookups = Some(Arc::clone(&l));
        Ok(l)
    }

    /// `PcRow::shared` (`bytecode_read_raf.rs:92-150`): the PC column the address phase pushes forward along (unmapped rows on 0) and the committed chunks of the mapped PC
    /// (unmapped rows cold) the cycle phase folds.
    pub fn pc(session: &mut ProofSession, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, log_t: usize, log_k: usize, chunk_bits: usize) -> Result<Arc<ResidentPc>, KernelError<Fr>> {
        if let Some(p) = &Self::entry(session).pc {
            return Ok(Arc::clone(p));
        }
        let cycles = 1usize << log_t;
        let bundles: Vec<PcBundle> = collect_bundles(witness, cycles).map_err(witness_error)?;
        let push: Vec<u64> = bundles.iter().map(|b| b.bytecode_pc.0 as u64).collect();
        let n_chunks = log_k.div_ceil(chunk_bits);
        let mask = (1u64 << chunk_bits) - 1;
        let mut hot = vec![0xFFu8; n_chunks * cycles];
        for (j, b) in bundles.iter().enumerate() {
            if let Some(pc) = b.mapped_pc.0 {
                for i in 0..n_chunks {
                    hot[i * cycles + j] = ((pc as u64 >> ((n_chunks - 1 - i) * chunk_bits)) & mask) as u8;
                }
            }
        }
        let p = Arc::new(ResidentPc {
            first_pc: push.first().copied().unwrap_or(0),
            push_pc: HipInts::from_u64(ctx, &push)?,
            chunks: HipHotIndices::upload(ctx, &hot, n_chunks, cycles, 1u32 << chunk_bits)?,
            chunk_bits: chunk_bits as u32,
        });
        Self::entry(session).pc = Some(Arc::clone(&p));
        Ok(p)
    }

    /// Every committed RA chunk column in the layout's canonical order (instruction, bytecode, RAM): `ColumnSelector::for_layout` (`optimized/booleanity.rs`) /
    /// `FamilySelectors` (`hamming_weight_claim_reduction.rs:83-117`) as one-byte hot indices, cold where the source has no address.
    pub fn ra_columns(
        session: &mut ProofSession,
        ctx: &Arc<HipContext>,
        witness: &dyn JoltWitnessPlane<Fr>,
        log_t: usize,
        log_k_chunk: usize,
        counts: (usize, usize, usize),
    ) -> Result<Arc<HipHotIndices>, KernelError<Fr>> {
        if let Some(c) = &Self::entry(session).ra_columns {
            return Ok(Arc::clone(c));
        }
        let cycles = 1usize << log_t;
        let lookups: Vec<LookupBundle> = collect_bundles(witness, cycles).map_err(witness_error)?;
        let pcs: Vec<PcBundle> = collect_bundles(witness, cycles).map_err(witness_error)?;
        let rams: Vec<RamAccessBundle> = collect_bundles(witness, cycles).map_err(witness_error)?;
        let (n_ins, n_bc, n_ram) = counts;
        let total = n_ins + n_bc + n_ram;
        let mut hot = vec![0xFFu8; total * cycles];
        let select = |index: usize, count: usize| RaChunkSelector::new(index, count, log_k_chunk).map_err(KernelError::from);
        for i in 0..n_ins {
            let s = select(i, n_ins)?;
            for j in 0..cycles {
                hot[i * cycles + j] = s.chunk_u128(lookups[j].lookup_index.0) as u8;
            }
        }
        for i in 0..n_bc {
            let s = select(i, n_bc)?;
            for j in 0..cycles {
                if let Some(pc) = pcs[j].mapped_pc.0 {
                    hot[(n_ins + i) * cycles + j] = s.chunk_u128(pc as u128) as u8;
                }
            }
        }
        for i in 0..n_ram {
            let s = select(i, n_ram)?;
            for j in 0..cycles {
                if let Some(a) = rams[j].address.0 {
                    hot[(n_ins + n_bc + i) * cycles + j] = s.chunk_u128(u128::from(a)) as u8;
                }
            }
        }
        let c = Arc::new(HipHotIndices::upload(ctx, &hot, total, cycles, 1u32 << log_k_chunk)?);
        Self::entry(session).ra_columns = Some(Arc::clone(&c));
        Ok(c)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// the slots
// ------------------------------------------------------------------------------------------------------------------
macro_rules! slot {
    ($(#[$doc:meta])* $name:ident) => {
        $(#[$doc])*
        pub struct $name {
            pub ctx: Arc<HipContext>,
        }
        impl $name {
            pub fn new(ctx: &Arc<HipContext>) -> Self {
                Self { ctx: Arc::clone(ctx) }
            }
        }
    };
}

slot!(
    /// `backend.ram_read_write` (replaces `optimized/ram_read_write.rs:269-335`).
    HipRamReadWrite
);
impl PrepareKernel<Fr, RamReadWriteChecking<Fr>> for HipRamReadWrite {
    #[tracing::instrument(skip_all, name = "HipRamReadWrite::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, RamReadWriteChecking<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = RamReadWriteChecking<Fr>>>, KernelError<Fr>> {
        let relation = inputs.relation;
        let dimensions = relation.dimensions();
        let (log_t, log_k, tau_low) = (dimensions.log_t(), relation.ram_log_k(), relation.product_tau_low());
        if dimensions.phase1_num_rounds() != log_t {
            return Err(KernelError::Unsupported { reason: "the device RAM read-write kernel supports only the default read-write config (phase 1 = all cycle rounds)" });
        }
        if log_t == 0 || dimensions.log_k() != log_k || tau_low.len() != log_t {
            return Err(KernelError::InvariantViolation { reason: "RAM read-write checking geometry is inconsistent" });
        }
        let ram = ResidentTrace::ram(session, &self.ctx, witness, log_t, log_k)?;
        let gamma = inputs.challenges.gamma;
        let mut raw = ptr::null_mut();
        // SAFETY: live handles of T (columns, inc) and K (val_init) entries; tau_low holds log_t elements, gamma one, of jolt_fr_t layout.
        check(
            unsafe {
                ffi::jolt_stage_ram_read_write_create(
                    self.ctx.raw,
                    ram.addresses.raw,
                    ram.pre_values.raw,
                    ram.post_values.raw,
                    ram.inc.raw,
                    ram.val_init.raw,
                    tau_low.as_ptr().cast(),
                    (&gamma as *const Fr).cast(),
                    &mut raw,
                )
            },
            self.ctx.raw,
        )?;
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        // output values: {ra, val, inc, bound cycle-eq factor} (jolt_stage_ram_read_write_create)
        Ok(kernel(op, vec![ram], Box::new(|v, _| {
            short(v, 3)?;
            Ok(RamReadWriteOutputClaims { val: v[1], ra: v[0], inc: v[2] })
        })))
    }
}

slot!(
    /// `backend.registers_read_write` (replaces `optimized/registers_read_write/mod.rs:81-215`).
    HipRegistersReadWrite
);
impl PrepareKernel<Fr, RegistersReadWriteChecking<Fr>> for HipRegistersReadWrite {
    #[tracing::instrument(skip_all, name = "HipRegistersReadWrite::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, RegistersReadWriteChecking<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = RegistersReadWriteChecking<Fr>>>, KernelError<Fr>> {
        let dimensions = inputs.relation.register_dimensions();
        if dimensions.phase1_num_rounds() != dimensions.log_t() {
            return Err(KernelError::Unsupported { reason: "the device registers read-write kernel supports only the default read-write config (phase 1 = all cycle rounds)" });
        }
        let log_t = dimensions.log_t();
        if log_t == 0 || dimensions.log_k() != REGISTER_ADDRESS_BITS {
            return Err(KernelError::Unsupported { reason: "the device registers read-write kernel needs at least one cycle round and the 7-bit register domain" });
        }
        let r_cycle: &[Fr] = &inputs.points.rd_write_value;
        if r_cycle.len() != log_t {
            return Err(KernelError::InvariantViolation { reason: "registers read-write input point has the wrong variable count" });
        }
        let regs = ResidentTrace::registers(session, &self.ctx, witness, log_t)?;
        let gamma = inputs.challenges.gamma;
        let mut raw = ptr::null_mut();
        // SAFETY: live handles over T cycles; r_cycle holds log_t elements, gamma one.
        check(
            unsafe {
                ffi::jolt_stage_registers_read_write_create(
                    self.ctx.raw,
                    regs.indices.raw,
                    regs.rs1_val.raw,
                    regs.rs2_val.raw,
                    regs.rd_pre.raw,
                    regs.rd_post.raw,
                    regs.rd_inc.raw,
                    r_cycle.as_ptr().cast(),
                    (&gamma as *const Fr).cast(),
                    &mut raw,
                )
            },
            self.ctx.raw,
        )?;
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        // output values: {registers_val, rd_wa, gamma rs1_ra + gamma^2 rs2_ra, rd_inc, bound cycle-eq factor, rs1_ra, rs2_ra}
        Ok(kernel(op, vec![regs], Box::new(|v, _| {
            short(v, 7)?;
            Ok(RegistersReadWriteOutputClaims { registers_val: v[0], rs1_ra: v[5], rs2_ra: v[6], rd_wa: v[1], rd_inc: v[3] })
        })))
    }
}

slot!(
    /// `backend.ram_raf_evaluation` (replaces `optimized/ram_raf_evaluation.rs:29-82`).
    HipRamRafEvaluation
);
impl PrepareKernel<Fr, RamRafEvaluation<Fr>> for HipRamRafEvaluation {
    #[tracing::instrument(skip_all, name = "HipRamRafEvaluation::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, RamRafEvaluation<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = RamRafEvaluation<Fr>>>, KernelError<Fr>> {
        let relation = inputs.relation;
        let dimensions = relation.read_write_dimensions();
        let (ram_log_k, lowest_address, tau_low) = (relation.ram_log_k(), relation.lowest_address(), relation.tau_low());
        if dimensions.raf_evaluation_rounds() != ram_log_k {
            return Err(KernelError::Unsupported { reason: "the device RAM RAF evaluation supports only the default read-write config (phase 1 = all cycle rounds)" });
        }
        if tau_low.len() != dimensions.log_t() {
            return Err(KernelError::InvariantViolation { reason: "RAM RAF evaluation tau_low disagrees with the trace geometry" });
        }
        let ram = ResidentTrace::ram(session, &self.ctx, witness, dimensions.log_t(), ram_log_k)?;
        let index = Arc::new(HipKeyIndex::new(&self.ctx, &ram.addresses, 1u64 << ram_log_k)?);
        let mut raw = ptr::null_mut();
        // SAFETY: live index over T cycles; tau_low holds log_t elements.
        check(
            unsafe { ffi::jolt_stage_ram_raf_evaluation_create(self.ctx.raw, index.raw, tau_low.as_ptr().cast(), tau_low.len(), lowest_address, &mut raw) },
            self.ctx.raw,
        )?;
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        // output values: {ra_folded, unmap} bound
        Ok(kernel(op, vec![ram, index], Box::new(|v, _| {
            short(v, 1)?;
            Ok(RamRafEvaluationOutputClaims { ram_ra: v[0] })
        })))
    }
}

slot!(
    /// `backend.ram_output_check` (replaces `optimized/ram_output_check.rs:49-100`).
    HipRamOutputCheck
);
impl PrepareKernel<Fr, RamOutputCheck<Fr>> for HipRamOutputCheck {
    #[tracing::instrument(skip_all, name = "HipRamOutputCheck::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, RamOutputCheck<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = RamOutputCheck<Fr>>>, KernelError<Fr>> {
        let relation = inputs.relation;
        let output_address = inputs.challenges.output_address.as_slice();
        let ram_log_k = output_address.len();
        let dimensions = relation.read_write_dimensions();
        if dimensions.output_check_rounds() != ram_log_k {
            return Err(KernelError::Unsupported { reason: "the device RAM output check supports only the default read-write config (phase 1 = all cycle rounds)" });
        }
        // the public-IO words and the contiguous IO mask, exactly as the reference builds them (ram_output_check.rs:66-90)
        let public_memory = relation.public_memory();
        let addresses = 1usize << ram_log_k;
        let mut val_io = vec![0u64; addresses];
        for segment in &public_memory.segments {
            for (offset, &word) in segment.words.iter().enumerate() {
                let index = segment.start_index as usize + offset;
                if index < addresses {
                    val_io[index] = word;
                }
            }
        }
        let io_lo = u64::try_from(public_memory.io_mask_start.min(addresses as u128)).unwrap_or(addresses as u64);
        let io_hi = u64::try_from(public_memory.io_mask_end.min(addresses as u128)).unwrap_or(addresses as u64);
        let ram = ResidentTrace::ram(session, &self.ctx, witness, dimensions.log_t(), ram_log_k)?;
        let index = Arc::new(HipKeyIndex::new(&self.ctx, &ram.addresses, 1u64 << ram_log_k)?);
        let mut raw = ptr::null_mut();
        // SAFETY: live handles; val_init / val_io hold K words; output_address holds log K elements.
        check(
            unsafe {
                ffi::jolt_stage_ram_output_check_create(
                    self.ctx.raw,
                    index.raw,
                    ram.post_values.raw,
                    ram.val_init_host.as_ptr(),
                    val_io.as_ptr(),
                    io_lo,
                    io_hi.saturating_sub(io_lo),
                    output_address.as_ptr().cast(),
                    &mut raw,
                )
            },
            self.ctx.raw,
        )?;
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        Ok(kernel(op, vec![ram, index], Box::new(|v, _| {
            short(v, 1)?;
            Ok(RamOutputCheckOutputClaims { val_final: v[0] })
        })))
    }
}

slot!(
    /// `backend.booleanity_address` (replaces `optimized/booleanity.rs:243-283`).
    HipBooleanityAddress
);
impl PrepareKernel<Fr, BooleanityAddressPhase<Fr>> for HipBooleanityAddress {
    #[tracing::instrument(skip_all, name = "HipBooleanityAddress::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, BooleanityAddressPhase<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = BooleanityAddressPhase<Fr>>>, KernelError<Fr>> {
        let relation = inputs.relation;
        let dimensions = relation.dimensions();
        let (reference_address, gamma) = (inputs.challenges.reference_address.as_slice(), inputs.challenges.gamma);
        let reference_cycle = relation.reference_cycle();
        if reference_address.len() != dimensions.log_k_chunk || reference_cycle.len() != dimensions.log_t {
            return Err(KernelError::InvariantViolation { reason: "booleanity reference point lengths disagree with the dimensions" });
        }
        let layout = dimensions.layout;
        let columns = ResidentTrace::ra_columns(session, &self.ctx, witness, dimensions.log_t, dimensions.log_k_chunk, (layout.instruction(), layout.bytecode(), layout.ram()))?;
        let mut raw = ptr::null_mut();
        // SAFETY: live columns over T cycles; the points hold log T / log K elements, gamma one.
        check(
            unsafe {
                ffi::jolt_stage_booleanity_address_create(
                    self.ctx.raw,
                    columns.raw,
                    reference_cycle.as_ptr().cast(),
                    reference_cycle.len(),
                    reference_address.as_ptr().cast(),
                    (&gamma as *const Fr).cast(),
                    &mut raw,
                )
            },
            self.ctx.raw,
        )?;
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        Ok(kernel(op, vec![columns], Box::new(|v, _| {
            short(v, 1)?;
            Ok(BooleanityAddressPhaseOutputClaims { intermediate: v[0] })
        })))
    }
}

slot!(
    /// `backend.booleanity_cycle` (replaces `optimized/booleanity.rs:436-690`): the stage-6b cycle phase over the same resident RA columns, lazily bound.
    HipBooleanityCycle
);
impl PrepareKernel<Fr, Booleanity<Fr>> for HipBooleanityCycle {
    #[tracing::instrument(skip_all, name = "HipBooleanityCycle::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, Booleanity<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = Booleanity<Fr>>>, KernelError<Fr>> {
        let relation = inputs.relation;
        let dimensions = relation.dimensions();
        let layout = dimensions.layout;
        let (r_address, reference_address, reference_cycle) = (relation.r_address(), relation.reference_address(), relation.reference_cycle());
        if r_address.len() != dimensions.log_k_chunk || reference_address.len() != dimensions.log_k_chunk || reference_cycle.len() != dimensions.log_t {
            return Err(KernelError::InvariantViolation { reason: "booleanity cycle-phase point lengths disagree with the dimensions" });
        }
        // the operator serves exactly the base RA layout's members (optimized/booleanity.rs:459-481 fails closed on the lattice variant the same way)
        let openings: Vec<JoltOpeningId> = layout.openings(JoltRelationId::Booleanity).collect();
        if openings.len() != layout.total() {
            return Err(KernelError::Unsupported { reason: "the device booleanity cycle operator serves the base RA layout only" });
        }
        let columns = ResidentTrace::ra_columns(session, &self.ctx, witness, dimensions.log_t, dimensions.log_k_chunk, (layout.instruction(), layout.bytecode(), layout.ram()))?;
        let gamma = inputs.challenges.gamma;
        let mut raw = ptr::null_mut();
        // SAFETY: live columns over T cycles; r_address / reference_address hold log K elements, reference_cycle log T, gamma one.
        check(
            unsafe {
                ffi::jolt_stage_booleanity_cycle_create(
                    self.ctx.raw,
                    columns.raw,
                    r_address.as_ptr().cast(),
                    reference_address.as_ptr().cast(),
                    reference_cycle.as_ptr().cast(),
                    reference_cycle.len(),
                    (&gamma as *const Fr).cast(),
                    &mut raw,
                )
            },
            self.ctx.raw,
        )?;
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        // output values: the bound columns, unscaled by gamma^-i inside the operator, in the layout's canonical order (booleanity.rs:652-662)
        Ok(kernel(op, vec![columns], Box::new(move |v, _| {
            short(v, openings.len())?;
            SumcheckOutputClaims::<Fr, Booleanity<Fr>>::from_opening_values(|id: &JoltOpeningId| openings.iter().position(|o| o == id).map(|k| v[k])).map_err(SumcheckKernelError::from)
        })))
    }
}

slot!(
    /// `backend.hamming_weight_claim_reduction` (replaces `optimized/hamming_weight_claim_reduction.rs:149-240`).
    HipHammingWeightClaimReduction
);
impl PrepareKernel<Fr, HammingWeightClaimReduction<Fr>> for HipHammingWeightClaimReduction {
    #[tracing::instrument(skip_all, name = "HipHammingWeightClaimReduction::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, HammingWeightClaimReduction<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = HammingWeightClaimReduction<Fr>>>, KernelError<Fr>> {
        let relation = inputs.relation;
        let dimensions = relation.dimensions();
        let layout = dimensions.layout;
        let (r_cycle, r_address, virtualization_points) = (relation.r_cycle(), relation.r_address(), relation.virtualization_points());
        if r_address.len() != dimensions.log_k_chunk || virtualization_points.len() != layout.total() || virtualization_points.iter().any(|p| p.len() != dimensions.log_k_chunk) {
            return Err(KernelError::InvariantViolation { reason: "hamming reduction reference point shapes disagree with the layout" });
        }
        let columns = ResidentTrace::ra_columns(session, &self.ctx, witness, r_cycle.len(), dimensions.log_k_chunk, (layout.instruction(), layout.bytecode(), layout.ram()))?;
        let flat: Vec<Fr> = virtualization_points.iter().flat_map(|p| p.iter().copied()).collect();
        let gamma = inputs.challenges.gamma;
        let mut raw = ptr::null_mut();
        // SAFETY: live columns; r_cycle log T elements, r_address log K, flat = n_polys x log K, gamma one.
        check(
            unsafe {
                ffi::jolt_stage_hamming_weight_create(
                    self.ctx.raw,
                    columns.raw,
                    r_cycle.as_ptr().cast(),
                    r_cycle.len(),
                    r_address.as_ptr().cast(),
                    flat.as_ptr().cast(),
                    (&gamma as *const Fr).cast(),
                    &mut raw,
                )
            },
            self.ctx.raw,
        )?;
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        // output values: the bound G_i in the layout's canonical order = layout.openings(HammingWeightClaimReduction) (hamming_weight_claim_reduction.rs:212-215)
        let openings: Vec<JoltOpeningId> = layout.openings(JoltRelationId::HammingWeightClaimReduction).collect();
        Ok(kernel(op, vec![columns], Box::new(move |v, _| {
            short(v, openings.len())?;
            SumcheckOutputClaims::<Fr, HammingWeightClaimReduction<Fr>>::from_opening_values(|id: &JoltOpeningId| openings.iter().position(|o| o == id).map(|k| v[k]))
                .map_err(SumcheckKernelError::from)
        })))
    }
}

slot!(
    /// `backend.instruction_read_raf` (replaces `optimized/instruction_read_raf.rs:351-430`): 128 address rounds + log T cycle rounds in ONE operator.
    HipInstructionReadRaf
);
impl PrepareKernel<Fr, InstructionReadRaf<Fr>> for HipInstructionReadRaf {
    #[tracing::instrument(skip_all, name = "HipInstructionReadRaf::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, InstructionReadRaf<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = InstructionReadRaf<Fr>>>, KernelError<Fr>> {
        let dimensions = inputs.relation.dimensions();
        // the reduction point is the consumed lookup-output opening point (OptimizedInstructionReadRafKernel::new(dimensions, &inputs.points.lookup_output, ..), :375-380)
        let r_reduction: &[Fr] = &inputs.points.lookup_output;
        let log_t = dimensions.log_t();
        if r_reduction.len() != log_t {
            return Err(KernelError::InvariantViolation { reason: "instruction read-RAF reduction point has the wrong variable count" });
        }
        let lookups = ResidentTrace::lookups(session, &self.ctx, witness, log_t)?;
        // a flag claim for every lookup table, in LookupTableKind order (output_claims :1390-1420: num_tables = LookupTableKind::COUNT = 42 at XLEN = 64)
        let present = [1u8; 42];
        let n_present = present.len();
        let ra_count = dimensions.num_virtual_ra_polys();
        let gamma = inputs.challenges.gamma;
        let mut raw = ptr::null_mut();
        // SAFETY: live rows / claim columns over T cycles; r_reduction log T elements, gamma one, present 42 bytes.
        check(
            unsafe {
                ffi::jolt_stage_instruction_read_raf_create(
                    self.ctx.raw,
                    lookups.rows.raw,
                    lookups.claim_columns.raw,
                    r_reduction.as_ptr().cast(),
                    log_t,
                    (&gamma as *const Fr).cast(),
                    present.as_ptr(),
                    ra_count as u32,
                    &mut raw,
                )
            },
            self.ctx.raw,
        )?;
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        // output values: {lookup_table_flags of the present tables, instruction_raf_flag, the ra_count bound ra_i}
        Ok(kernel(op, vec![lookups], Box::new(move |v, _| {
            short(v, n_present + 1 + ra_count)?;
            Ok(InstructionReadRafOutputClaims {
                lookup_table_flags: v[..n_present].to_vec(),
                instruction_ra: v[n_present + 1..n_present + 1 + ra_count].to_vec(),
                instruction_raf_flag: v[n_present],
            })
        })))
    }
}

/// What the bytecode address phase parks for the cycle phase (`SumcheckKernel::park_residue`): the finished operator, whose eq tables and bound values the cycle
/// operator is prepared from (`jolt_stage_bytecode_read_raf_cycle_create`).
pub struct HipBytecodeAddressResidue(pub HipStageOp);
crate::status::zero_host_heap!(HipBytecodeAddressResidue);

slot!(
    /// `backend.bytecode_read_raf_address` (replaces `optimized/bytecode_read_raf.rs:241-330`).  `stage_values`: the K-sized per-stage value tables of the program
    /// (`read_raf_stage_values`, O(K) host work from the program image: crate-private in `jolt-kernels` today, handed in by the constructor's caller).
    HipBytecodeReadRafAddress
);
/// `read_raf_stage_values` over the relation's points and the drawn stage gammas: `[stage][k]`, 5 x K field elements.
pub type StageValues = fn(&BytecodeReadRafAddressPhase<Fr>, &ConcreteSumcheckChallenges<Fr, BytecodeReadRafAddressPhase<Fr>>, &dyn JoltWitnessPlane<Fr>) -> Result<Vec<Fr>, KernelError<Fr>>;
pub struct HipBytecodeReadRafAddressWith {
    pub slot: HipBytecodeReadRafAddress,
    pub stage_values: StageValues,
}
impl PrepareKernel<Fr, BytecodeReadRafAddressPhase<Fr>> for HipBytecodeReadRafAddressWith {
    #[tracing::instrument(skip_all, name = "HipBytecodeReadRafAddress::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, BytecodeReadRafAddressPhase<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = BytecodeReadRafAddressPhase<Fr>>>, KernelError<Fr>> {
        let ctx = &self.slot.ctx;
        let relation = inputs.relation;
        let dimensions = relation.dimensions();
        let (log_t, log_k) = (dimensions.log_t(), dimensions.log_k());
        let stage_values = (self.stage_values)(relation, inputs.challenges, witness)?;
        if stage_values.len() != 5usize << log_k {
            return Err(KernelError::TableSizeMismatch { table: "bytecode stage values".to_owned(), expected: 5usize << log_k, got: stage_values.len() });
        }
        let stage_cycle_points = relation.stage_cycle_points();
        if stage_cycle_points.len() != 5 || stage_cycle_points.iter().any(|p| p.len() != log_t) {
            return Err(KernelError::InvariantViolation { reason: "bytecode stage cycle point has the wrong variable count" });
        }
        let flat: Vec<Fr> = stage_cycle_points.iter().flat_map(|p| p.iter().copied()).collect();
        // (the chunk columns ride along for the cycle phase: d committed chunks of ceil(log K / d) bits, BytecodeReadRafCycle::committed_chunk_bits)
        let pc = ResidentTrace::pc(session, ctx, witness, log_t, log_k, log_k.div_ceil(dimensions.num_committed_ra_polys().max(1)))?;
        let index = Arc::new(HipKeyIndex::new(ctx, &pc.push_pc, 1u64 << log_k)?);
        let gamma = inputs.challenges.gamma;
        let committed_program = relation.committed_program();
        let mut raw = ptr::null_mut();
        // SAFETY: live index over T cycles; flat = 5 x log T elements, stage_values 5 x K, gamma one.
        check(
            unsafe {
                ffi::jolt_stage_bytecode_read_raf_address_create(
                    ctx.raw,
                    index.raw,
                    flat.as_ptr().cast(),
                    log_t,
                    stage_values.as_ptr().cast(),
                    (&gamma as *const Fr).cast(),
                    pc.first_pc,
                    relation.entry_bytecode_index() as u64,
                    &mut raw,
                )
            },
            ctx.raw,
        )?;
        let op = HipStageOp::adopt(ctx, raw)?;
        // output values: the 13 bound tables (F_0..4, V_0..4, Int, entry_trace, entry_expected), then the intermediate claim
        let mut k = HipStageKernel {
            op,
            extract: Box::new(move |v: &[Fr], _: &SumcheckInputClaims<Fr, BytecodeReadRafAddressPhase<Fr>>| {
                short(v, 14)?;
                Ok(BytecodeReadRafAddressPhaseOutputClaims { intermediate: v[13], val_stages: if committed_program { v[5..10].to_vec() } else { Vec::new() } })
            }),
            park: None,
            _keep: vec![pc, index],
        };
        k.park = Some(|op, session| session.park(HipBytecodeAddressResidue(op)));
        Ok(Box::new(k))
    }
}

slot!(
    /// `backend.bytecode_read_raf_cycle` (replaces `optimized/bytecode_read_raf.rs:464-560`): prepared from the address phase's parked operator.
    HipBytecodeReadRafCycle
);
impl PrepareKernel<Fr, BytecodeReadRafCycle<Fr>> for HipBytecodeReadRafCycle {
    #[tracing::instrument(skip_all, name = "HipBytecodeReadRafCycle::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, BytecodeReadRafCycle<Fr>>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = BytecodeReadRafCycle<Fr>>>, KernelError<Fr>> {
        let relation = inputs.relation;
        let dimensions = relation.dimensions();
        let chunk_bits = relation.committed_chunk_bits();
        let chunks = committed_address_chunks(relation.r_address(), chunk_bits);
        let num_ra = dimensions.num_committed_ra_polys();
        if chunks.len() != num_ra {
            return Err(KernelError::InvariantViolation { reason: "bytecode address chunk count disagrees with the committed RA count" });
        }
        let HipBytecodeAddressResidue(address) =
            session.take::<HipBytecodeAddressResidue>().ok_or(KernelError::InvariantViolation { reason: "stage 6a parked no bytecode address operator for the cycle phase" })?;
        let pc = ResidentTrace::pc(session, &self.ctx, witness, dimensions.log_t(), dimensions.log_k(), chunk_bits)?;
        let mut raw = ptr::null_mut();
        // SAFETY: the address operator has finished its rounds (the stage driver parks residues after the round loop); live chunk columns over T cycles.
        check(unsafe { ffi::jolt_stage_bytecode_read_raf_cycle_create(self.ctx.raw, address.raw, pc.chunks.raw, pc.chunk_bits, &mut raw) }, self.ctx.raw)?;
        drop(address);
        let op = HipStageOp::adopt(&self.ctx, raw)?;
        // output values: the bound ra_i in chunk order = read_raf_output_openings(dimensions).bytecode_ra (bytecode_read_raf.rs:536-545)
        let openings = jolt_claims::protocols::jolt::geometry::bytecode::read_raf_output_openings(dimensions).bytecode_ra;
        Ok(kernel(op, vec![pc], Box::new(move |v, _| {
            short(v, openings.len())?;
            SumcheckOutputClaims::<Fr, BytecodeReadRafCycle<Fr>>::from_opening_values(|id: &JoltOpeningId| openings.iter().position(|o| o == id).map(|k| v[k]))
                .map_err(SumcheckKernelError::from)
        })))
    }
}

/// Field weights of the remainder member at the uni-skip challenge: `[stream][1 + inputs]` for A and for B (the Lagrange kernel at r0 folded over the constraint rows,
/// `optimized/spartan_outer.rs:236-300`; crate-private helpers of `jolt-kernels`, handed in like [`crate::backend::NodeWeights`]) and the member's scale.
pub type RemainderWeights<R> = fn(&R, &HipUniskipCarry) -> Result<(Vec<Fr>, Vec<Fr>, Fr), KernelError<Fr>>;

/// `backend.spartan_outer_remainder` / `backend.spartan_product_remainder` (replace `optimized/spartan_outer.rs:525-700`, `spartan_product.rs:278-437`): the columns
/// the uni-skip front left resident ([`HipUniskipCarry`]) feed the remainder's Az / Bz; the claimed inputs come from the same columns.
pub struct HipSpartanRemainder<R> {
    pub ctx: Arc<HipContext>,
    pub weights: RemainderWeights<R>,
    /// output opening ids in the carry's column order: value k is the claimed input of column k
    pub openings: fn(&R) -> Vec<JoltOpeningId>,
}

macro_rules! impl_remainder {
    ($relation:ty) => {
        impl PrepareKernel<Fr, $relation> for HipSpartanRemainder<$relation> {
            #[tracing::instrument(skip_all, name = "HipSpartanRemainder::prepare")]
            fn prepare(
                &self,
                session: &mut ProofSession,
                _witness: &dyn JoltWitnessPlane<Fr>,
                inputs: ProverInputs<'_, Fr, $relation>,
            ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = $relation>>, KernelError<Fr>> {
                let carry = session.take::<HipUniskipCarry>().ok_or(KernelError::InvariantViolation { reason: "the uni-skip slot parked no carry for the remainder member" })?;
                let rounds = inputs.relation.rounds();
                if rounds != carry.log_t + carry.streams as usize - 1 || carry.tau.len() < rounds {
                    return Err(KernelError::InvariantViolation { reason: "remainder rounds disagree with the uni-skip carry" });
                }
                let (a, b, scale) = (self.weights)(inputs.relation, &carry)?;
                let width = carry.streams as usize * (1 + carry.columns.len());
                if a.len() != width || b.len() != width {
                    return Err(KernelError::InvariantViolation { reason: "remainder column weights disagree with the resident columns" });
                }
                let columns: Vec<*const ffi::jolt_ints> = carry.columns.iter().map(|c| c.raw.cast_const()).collect();
                let mut raw = ptr::null_mut();
                // SAFETY: live integer columns over T cycles; weights = streams x (1 + n) elements each; tau holds `rounds` elements; scale one.
                check(
                    unsafe {
                        ffi::jolt_stage_spartan_remainder_create(
                            self.ctx.raw,
                            columns.as_ptr(),
                            columns.len(),
                            carry.streams,
                            a.as_ptr().cast(),
                            b.as_ptr().cast(),
                            carry.tau.as_ptr().cast(),
                            rounds,
                            (&scale as *const Fr).cast(),
                            &mut raw,
                        )
                    },
                    self.ctx.raw,
                )?;
                let op = HipStageOp::adopt(&self.ctx, raw)?;
                let openings = (self.openings)(inputs.relation);
                let keep: Arc<dyn core::any::Any + Send + Sync> = Arc::new(ResidentColumns(carry.columns));
                Ok(kernel(op, vec![keep], Box::new(move |v, inputs| {
                    short(v, openings.len())?;
                    SumcheckOutputClaims::<Fr, $relation>::from_opening_values(|id: &JoltOpeningId| openings.iter().position(|o| o == id).map(|k| v[k]).or_else(|| inputs.resolve_input(id)))
                        .map_err(SumcheckKernelError::from)
                })))
            }
        }
    };
}
impl_remainder!(OuterRemainder<Fr>);
impl_remainder!(ProductRemainder<Fr>);

/// The uni-skip carry's columns kept alive under the remainder operator that borrows them.
struct ResidentColumns(Vec<HipInts>);
// SAFETY: device handles owned by the value, used from one thread at a time (see HipContext).
unsafe impl Sync for ResidentColumns {}
