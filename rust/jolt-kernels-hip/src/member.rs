//! Sumcheck members on the device: the twin of `NaiveSumcheckProver` (`crates/jolt-kernels/src/reference/naive.rs:53-377`) and
//! of the optimized tier's split-eq members, behind `ProveRounds` / `SumcheckKernel` / `PrepareKernel`.
use std::cell::RefCell;
use std::collections::BTreeMap;
use std::ptr;
use std::rc::Rc;
use std::sync::Arc;

use jolt_claims::protocols::jolt::{JoltChallengeId, JoltDerivedId, JoltOpeningId};
use jolt_claims::{InputClaims, OutputClaims, Source, SumcheckChallenges, SymbolicSumcheck};
use jolt_field::Fr;
use jolt_kernels::{KernelError, PrepareKernel, ProofSession, ProverInputs, SumcheckKernel, SumcheckKernelError};
use jolt_poly::{BindingOrder, GruenSplitEqPolynomial, UnivariatePoly};
use jolt_sumcheck::{ProveRounds, SumcheckError};
use jolt_verifier::stages::relations::{ConcreteSumcheck, ConcreteSumcheckChallenges, SumcheckInputClaims, SumcheckOutputClaims};
use jolt_witness::JoltWitnessPlane;

use crate::context::{HipContext, HipTable};
use crate::ffi;
use crate::status::{check, to_kernel_seam_error, to_sumcheck_error, HipError};

/// What `jolt_member_prove_round` returns for a member, i.e. how the host completes the round polynomial.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum MemberShape {
    /// `degree + 1` evaluations `s(0..=degree)` (`jolt_member_create_expr`): `UnivariatePoly::from_evals`.
    Evals { degree: usize },
    /// `degree` evaluations at `{0, 2, .., degree}`; `s(1) = claim - s(0)` (`JOLT_MEMBER_FLAG_SKIP_ONE`,
    /// `crates/jolt-kernels/src/optimized/support.rs:450-459 round_poly_from_skipped_evals`).
    SkippedOne { degree: usize },
    /// `(q(0), q(inf))` of an `eq * a * b` member; the cubic is `GruenSplitEqPolynomial::gruen_poly_deg_3`
    /// (`crates/jolt-poly/src/split_eq.rs:383-417`) over the eq scalar state returned in `aux`.
    GruenProduct,
}

/// One slot of [`HipMember::new_lc`]: a field table, or a resident `u64` witness column read as compact scalars until the first bind.
pub enum MemberSlot<'a> {
    Table(&'a HipTable),
    Ints(&'a crate::ops::HipInts),
}

/// Round sums a [`crate::scheduler::HipRoundScheduler`] fetched for this member with one grouped launch; `prove_round` consumes them
/// instead of launching on its own.
pub(crate) type Mailbox = Rc<RefCell<Option<Vec<Fr>>>>;

/// One device-resident batch member: the object behind `ProveRounds`.
pub struct HipMember {
    pub(crate) ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_member,
    pub(crate) shape: MemberShape,
    rounds: usize,
    bound: usize,
    pub(crate) mailbox: Mailbox,
    /// Split-eq members: the reference's own eq state on the host (O(1) scalar update per bind; its cached sqrt(N) tables are never
    /// read here -- the device holds its own).  It completes the cubic from the two device sums (`gruen_poly_deg_3`).
    eq: Option<GruenSplitEqPolynomial<Fr>>,
}

impl HipMember {
    /// `NaiveSumcheckProver::new` (`naive.rs:136-205`) with the leaf tables already resident: `terms[k] = (coefficient, leaf table
    /// indices)`, challenge leaves folded into the coefficients.  Takes ownership of `tables`.
    pub fn new_expr(ctx: &Arc<HipContext>, tables: Vec<HipTable>, terms: &[(Fr, Vec<u32>)], degree: usize, order: BindingOrder) -> Result<Self, HipError> {
        let mut offsets = vec![0u32];
        let mut factors = Vec::new();
        let mut coeffs = Vec::with_capacity(terms.len());
        for (c, f) in terms {
            factors.extend_from_slice(f);
            offsets.push(factors.len() as u32);
            coeffs.push(*c);
        }
        if factors.is_empty() {
            factors.push(0);
        }
        let desc = ffi::jolt_member_desc {
            n_tables: tables.len() as u32,
            n_terms: terms.len() as u32,
            degree: degree as u32,
            order: match order {
                BindingOrder::LowToHigh => ffi::JOLT_ORDER_LOW_TO_HIGH,
                BindingOrder::HighToLow => ffi::JOLT_ORDER_HIGH_TO_LOW,
            },
            term_offsets: offsets.as_ptr(),
            factors: factors.as_ptr(),
            coeffs: coeffs.as_ptr().cast(),
        };
        let handles: Vec<*mut ffi::jolt_table> = tables.into_iter().map(HipTable::into_raw).collect();
        let mut raw = ptr::null_mut();
        // SAFETY: descriptor arrays outlive the call (the library copies them); the table handles are live and ownership moves
        // into the member on success (on failure the library leaves them untouched and they leak only if we return early here).
        check(unsafe { ffi::jolt_member_create_expr(ctx.raw, handles.as_ptr(), &desc, &mut raw) }, ctx.raw)?;
        let mut rounds = 0usize;
        // SAFETY: live member.
        check(unsafe { ffi::jolt_member_num_rounds(raw, &mut rounds) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw, shape: MemberShape::Evals { degree }, rounds, bound: 0, mailbox: Mailbox::default(), eq: None })
    }

    /// A member in the optimized tier's "sum of products of linear combinations" form over slots that are field tables OR compact-scalar witness columns
    /// (`Polynomial<T>` before its first bind, `crates/jolt-poly/src/dense.rs:129-142`): `groups[g]` is a product of factors, a factor is
    /// `(constant, [(coefficient, slot)])`.  Integer slots are read as they lie in round 0 and turned into field tables by the first bind
    /// (`jolt_member_create_lc_small`; `bind_to_field`); with field slots only this is `jolt_member_create_lc`.  The slots are BORROWED: they must outlive the
    /// member (the reference's members borrow the witness the same way).  `skip_one`: the member returns `s(0), s(2), .., s(degree)` and the host recovers `s(1)`
    /// from the claim (`round_poly_from_skipped_evals`, `crates/jolt-kernels/src/optimized/support.rs:450-459`).
    pub fn new_lc(
        ctx: &Arc<HipContext>,
        slots: &[MemberSlot<'_>],
        groups: &[Vec<(Option<Fr>, Vec<(Fr, u32)>)>],
        degree: usize,
        skip_one: bool,
    ) -> Result<Self, HipError> {
        let mut group_factor_offsets = vec![0u32];
        let mut factor_lc_offsets = vec![0u32];
        let (mut consts, mut lc_tables, mut lc_coeffs) = (Vec::new(), Vec::new(), Vec::new());
        for group in groups {
            for (constant, lc) in group {
                consts.push(constant.unwrap_or_default());
                for (coefficient, slot) in lc {
                    if *slot as usize >= slots.len() {
                        return Err(HipError::size_mismatch("a linear combination names a slot the member does not have"));
                    }
                    lc_coeffs.push(*coefficient);
                    lc_tables.push(*slot);
                }
                factor_lc_offsets.push(lc_tables.len() as u32);
            }
            group_factor_offsets.push(factor_lc_offsets.len() as u32 - 1);
        }
        let desc = ffi::jolt_member_lc_desc {
            n_tables: slots.len() as u32,
            n_groups: groups.len() as u32,
            n_factors: consts.len() as u32,
            n_lc: lc_tables.len() as u32,
            degree: degree as u32,
            order: ffi::JOLT_ORDER_LOW_TO_HIGH,
            flags: ffi::JOLT_MEMBER_FLAG_BORROW_TABLES | if skip_one { ffi::JOLT_MEMBER_FLAG_SKIP_ONE } else { 0 },
            group_factor_offsets: group_factor_offsets.as_ptr(),
            factor_lc_offsets: factor_lc_offsets.as_ptr(),
            factor_consts: consts.as_ptr().cast(),
            lc_tables: lc_tables.as_ptr(),
            lc_coeffs: lc_coeffs.as_ptr().cast(),
        };
        let tables: Vec<*mut ffi::jolt_table> = slots.iter().map(|s| match s { MemberSlot::Table(t) => t.raw, MemberSlot::Ints(_) => ptr::null_mut() }).collect();
        let ints: Vec<*const ffi::jolt_ints> = slots.iter().map(|s| match s { MemberSlot::Ints(v) => v.raw.cast_const(), MemberSlot::Table(_) => ptr::null() }).collect();
        let mut raw = ptr::null_mut();
        // SAFETY: the descriptor arrays outlive the call (the library copies them); exactly one of tables[i] / ints[i] is a live handle per slot; w = NULL selects the
        // plain (not eq-weighted) constructor, for which scale / shard_scale are unused.
        check(
            unsafe { ffi::jolt_member_create_lc_small(ctx.raw, tables.as_ptr(), ints.as_ptr(), &desc, ptr::null(), 0, ptr::null(), ptr::null(), &mut raw) },
            ctx.raw,
        )?;
        let mut rounds = 0usize;
        // SAFETY: live member.
        check(unsafe { ffi::jolt_member_num_rounds(raw, &mut rounds) }, ctx.raw)?;
        let shape = if skip_one { MemberShape::SkippedOne { degree } } else { MemberShape::Evals { degree } };
        Ok(Self { ctx: Arc::clone(ctx), raw, shape, rounds, bound: 0, mailbox: Mailbox::default(), eq: None })
    }

    /// `eq(w, j) * a(j) * b(j)` served from split-eq tables (`GruenSplitEqPolynomial::new_with_scaling(w, LowToHigh, scale)`,
    /// `crates/jolt-poly/src/split_eq.rs:187-236`); the canonical loop shape is `ram_hamming_booleanity.rs:111-135`.
    pub fn new_gruen_product(ctx: &Arc<HipContext>, a: HipTable, b: HipTable, w: &[Fr], scale: Option<Fr>) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        let scale_ptr = scale.as_ref().map_or(ptr::null(), |s| (s as *const Fr).cast());
        let (ra, rb) = (a.into_raw(), b.into_raw());
        // SAFETY: layouts of Fr / jolt_fr_t agree; handles live; ownership of a, b moves into the member.
        check(unsafe { ffi::jolt_member_create_split_eq_product(ctx.raw, ra, rb, w.as_ptr().cast(), w.len(), scale_ptr, &mut raw) }, ctx.raw)?;
        let eq = GruenSplitEqPolynomial::new_with_scaling(w, BindingOrder::LowToHigh, scale);
        Ok(Self { ctx: Arc::clone(ctx), raw, shape: MemberShape::GruenProduct, rounds: w.len(), bound: 0, mailbox: Mailbox::default(), eq: Some(eq) })
    }

    fn n_evals(&self) -> usize {
        match self.shape {
            MemberShape::Evals { degree } => degree + 1,
            MemberShape::SkippedOne { degree } => degree,
            MemberShape::GruenProduct => 2,
        }
    }

    /// The device half of one round: bind (if any) and fetch the round sums.
    fn round_sums(&mut self, bind: Option<Fr>) -> Result<Vec<Fr>, HipError> {
        if let (Some(r), Some(eq)) = (bind, self.eq.as_mut()) {
            eq.bind(r); // split_eq.rs:334-363, host half: current_scalar and current_index
        }
        if bind.is_some() {
            self.bound += 1;
        }
        if let Some(evals) = self.mailbox.borrow_mut().take() {
            return Ok(evals); // the stage's HipRoundScheduler already ran this member inside a grouped launch (bind included)
        }
        let mut evals = vec![Fr::default(); self.n_evals()];
        let bind_ptr = bind.as_ref().map_or(ptr::null(), |b| (b as *const Fr).cast());
        // SAFETY: buffers sized as the header prescribes for this member kind; layouts agree; aux_out may be null.
        check(unsafe { ffi::jolt_member_prove_round(self.raw, bind_ptr, evals.as_mut_ptr().cast(), evals.len(), ptr::null_mut()) }, self.ctx.raw)?;
        Ok(evals)
    }
}

impl ProveRounds<Fr> for HipMember {
    fn num_rounds(&self) -> usize {
        self.rounds
    }

    fn prove_round(&mut self, bind: Option<Fr>, round: usize, previous_claim: Fr) -> Result<UnivariatePoly<Fr>, SumcheckError<Fr>> {
        let evals = self.round_sums(bind).map_err(to_sumcheck_error)?;
        match self.shape {
            MemberShape::Evals { .. } => {
                // identical to the reference tier's tail (naive.rs:298-309)
                let round_sum = evals[0] + evals[1];
                if round_sum != previous_claim {
                    return Err(SumcheckError::RoundCheckFailed { round, expected: previous_claim, actual: round_sum });
                }
                Ok(UnivariatePoly::from_evals(&evals))
            }
            MemberShape::SkippedOne { .. } => {
                // support.rs:450-459: s(1) = claim - s(0), then interpolate on {0, 1, 2, .., degree}
                let mut full = Vec::with_capacity(evals.len() + 1);
                full.push(evals[0]);
                full.push(previous_claim - evals[0]);
                full.extend_from_slice(&evals[1..]);
                Ok(UnivariatePoly::from_evals(&full))
            }
            MemberShape::GruenProduct => match self.eq.as_ref() {
                // split_eq.rs:383-417: (q(0), q(inf), s(0) + s(1)) -> the cubic round polynomial
                Some(eq) => Ok(eq.gruen_poly_deg_3(evals[0], evals[1], previous_claim)),
                None => Err(SumcheckError::MissingEvaluationSource { kind: "derived" }),
            },
        }
    }

    fn finish_rounds(&mut self, bind: Fr) -> Result<(), SumcheckError<Fr>> {
        // SAFETY: live member; layouts agree.
        check(unsafe { ffi::jolt_member_finish(self.raw, (&bind as *const Fr).cast()) }, self.ctx.raw).map_err(to_sumcheck_error)?;
        if let Some(eq) = self.eq.as_mut() {
            eq.bind(bind);
        }
        self.bound += 1;
        Ok(())
    }
}

impl Drop for HipMember {
    fn drop(&mut self) {
        // SAFETY: owned handle (frees the tables the member owns).
        let _ = unsafe { ffi::jolt_member_destroy(self.raw) };
    }
}

/// The typed kernel: a [`HipMember`] plus the relation's leaf-id -> table-index map for `output_claims`.
pub struct HipSumcheckProver<R> {
    member: HipMember,
    relation: R,
    opening_index: BTreeMap<JoltOpeningId, usize>,
    n_tables: usize,
}

#[cfg(feature = "allocative")]
impl<R> allocative::Allocative for HipSumcheckProver<R> {
    fn visit<'a, 'b: 'a>(&self, visitor: &'a mut allocative::Visitor<'b>) {
        let mut visitor = visitor.enter_self_sized::<Self>();
        visitor.visit_simple(allocative::Key::new("heap"), 0usize); // the tables live in HBM
        visitor.exit();
    }
}

impl<R> ProveRounds<Fr> for HipSumcheckProver<R>
where
    R: ConcreteSumcheck<Fr>,
{
    fn num_rounds(&self) -> usize {
        self.member.num_rounds()
    }
    fn prove_round(&mut self, bind: Option<Fr>, round: usize, previous_claim: Fr) -> Result<UnivariatePoly<Fr>, SumcheckError<Fr>> {
        self.member.prove_round(bind, round, previous_claim)
    }
    fn finish_rounds(&mut self, bind: Fr) -> Result<(), SumcheckError<Fr>> {
        self.member.finish_rounds(bind)
    }
}

impl<R> SumcheckKernel<Fr> for HipSumcheckProver<R>
where
    R: ConcreteSumcheck<Fr>,
    SumcheckInputClaims<Fr, R>: InputClaims<Fr>,
    SumcheckOutputClaims<Fr, R>: OutputClaims<Fr>,
    ConcreteSumcheckChallenges<Fr, R>: SumcheckChallenges<Fr, JoltChallengeId>,
{
    type Relation = R;

    /// `naive.rs:331-347`: every opening table's fully bound value; ids no table serves are the dual-role openings read back
    /// from the consumed input claims.
    fn output_claims(&mut self, inputs: &SumcheckInputClaims<Fr, R>) -> Result<SumcheckOutputClaims<Fr, R>, SumcheckKernelError<Fr>> {
        let mut values = vec![Fr::default(); self.n_tables];
        let remaining = self.member.rounds - self.member.bound;
        // SAFETY: `values` holds n_tables elements of jolt_fr_t layout.
        check(unsafe { ffi::jolt_member_final_values(self.member.raw, values.as_mut_ptr().cast(), values.len()) }, self.member.ctx.raw)
            .map_err(|e| to_kernel_seam_error(e, remaining))?;
        let index = &self.opening_index;
        SumcheckOutputClaims::<Fr, R>::from_opening_values(|id| index.get(id).map(|&k| values[k]).or_else(|| inputs.resolve_input(id)))
            .map_err(SumcheckKernelError::from)
    }
}

/// `PrepareKernel` for any relation the reference tier serves with `NaiveSumcheckProver::new(&inputs, opening_tables,
/// derived_tables, order)`: the same constructor shape with the tables uploaded / expanded on the device.  `tables` resolves a
/// relation's leaves exactly as the reference slot does (`dense_view`, `eq_table`, `address_fold`, .. of
/// `crates/jolt-kernels/src/reference/views.rs:20-138`), but returns device tables.
pub struct HipPrepare<R, T> {
    pub ctx: Arc<HipContext>,
    pub order: BindingOrder,
    pub tables: T,
    _relation: core::marker::PhantomData<fn() -> R>,
}

/// Leaf resolution for one relation: opening leaves come from the witness plane, derived leaves from the stage's points.
pub trait ResolveLeaves<R: ConcreteSumcheck<Fr>>
where
    SumcheckInputClaims<Fr, R>: InputClaims<Fr>,
    SumcheckOutputClaims<Fr, R>: OutputClaims<Fr>,
    ConcreteSumcheckChallenges<Fr, R>: SumcheckChallenges<Fr, JoltChallengeId>,
{
    /// The table of an opening leaf: the witness column itself (`dense_view`), or its fold against a point of the relation (`address_fold`: the relation and its consumed
    /// points are in `inputs`).
    fn opening(&self, ctx: &Arc<HipContext>, witness: &dyn JoltWitnessPlane<Fr>, inputs: &ProverInputs<'_, Fr, R>, id: &JoltOpeningId) -> Result<HipTable, KernelError<Fr>>;
    fn derived(&self, ctx: &Arc<HipContext>, inputs: &ProverInputs<'_, Fr, R>, id: &JoltDerivedId) -> Result<HipTable, KernelError<Fr>>;
}

impl<R, T> HipPrepare<R, T> {
    pub fn new(ctx: Arc<HipContext>, order: BindingOrder, tables: T) -> Self {
        Self { ctx, order, tables, _relation: core::marker::PhantomData }
    }
}

impl<R, T> PrepareKernel<Fr, R> for HipPrepare<R, T>
where
    R: ConcreteSumcheck<Fr> + Clone + 'static,
    T: ResolveLeaves<R>,
    SumcheckInputClaims<Fr, R>: InputClaims<Fr>,
    SumcheckOutputClaims<Fr, R>: OutputClaims<Fr>,
    ConcreteSumcheckChallenges<Fr, R>: SumcheckChallenges<Fr, JoltChallengeId>,
{
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        inputs: ProverInputs<'_, Fr, R>,
    ) -> Result<Box<dyn SumcheckKernel<Fr, Relation = R>>, KernelError<Fr>> {
        let _span = tracing::info_span!("HipPrepare::prepare").entered();
        // The relation's output expression IS the summand (naive.rs:1-23): sum of `coefficient * prod(factors)` terms
        // (crates/jolt-claims/src/claims.rs:17-46).  Challenge factors fold into the coefficient (exact), opening and derived
        // factors become table indices.
        let expression = inputs.relation.symbolic().output_expression::<Fr>();
        let mut opening_index = BTreeMap::new();
        let mut derived_index: BTreeMap<JoltDerivedId, usize> = BTreeMap::new();
        let mut tables: Vec<HipTable> = Vec::new();
        let mut terms: Vec<(Fr, Vec<u32>)> = Vec::with_capacity(expression.terms.len());
        for term in &expression.terms {
            let mut coefficient = term.coefficient;
            let mut factors = Vec::with_capacity(term.factors.len());
            for factor in &term.factors {
                match factor {
                    Source::Challenge(id) => coefficient *= inputs.challenges.value(id).ok_or(KernelError::MissingChallenge { id: *id })?,
                    Source::Opening(id) => {
                        let k = match opening_index.get(id) {
                            Some(&k) => k,
                            None => {
                                tables.push(self.tables.opening(&self.ctx, witness, &inputs, id)?);
                                let _ = opening_index.insert(*id, tables.len() - 1);
                                tables.len() - 1
                            }
                        };
                        factors.push(k as u32);
                    }
                    Source::Derived(id) => {
                        let k = match derived_index.get(id) {
                            Some(&k) => k,
                            None => {
                                tables.push(self.tables.derived(&self.ctx, &inputs, id)?);
                                let _ = derived_index.insert(*id, tables.len() - 1);
                                tables.len() - 1
                            }
                        };
                        factors.push(k as u32);
                    }
                }
            }
            terms.push((coefficient, factors));
        }
        let n_tables = tables.len();
        let expected = 1usize << inputs.relation.rounds();
        for (k, t) in tables.iter().enumerate() {
            if t.len() != expected {
                return Err(KernelError::TableSizeMismatch { table: format!("leaf table {k}"), expected, got: t.len() });
            }
        }
        let member = HipMember::new_expr(&self.ctx, tables, &terms, inputs.relation.degree(), self.order)?;
        // register with the session so that the stage's round scheduler can launch all members of a round together
        crate::scheduler::register_member(session, &member);
        Ok(Box::new(HipSumcheckProver { member, relation: inputs.relation.clone(), opening_index, n_tables }))
    }
}
