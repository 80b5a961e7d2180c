//! `RoundScheduler` over device members: all members of one batch round go down in as few launches as possible
//! (`jolt_round_group_prove`), the engine's per-member `prove_round` calls then only assemble the round polynomials.
//!
//! The engine hands a scheduler `&mut dyn ProveRounds` handles (`MemberRound`, `crates/jolt-sumcheck/src/prover.rs:74-93`), which
//! cannot be downcast; device members therefore register `(raw handle, mailbox)` with the `ProofSession` during `prepare`
//! (the carry `BuildRoundScheduler::build` is given access to, `crates/jolt-kernels/src/backend.rs:64-70`) in batch order, and the
//! scheduler addresses them by `MemberRound::index`.
use std::ptr;
use std::rc::Rc;
use std::sync::{Arc, Mutex};

use jolt_field::Fr;
use jolt_kernels::{BuildRoundScheduler, ProofSession};
use jolt_sumcheck::{MemberFinish, MemberRound, RoundScheduler, SumcheckError};

use crate::context::HipContext;
use crate::ffi;
use crate::member::{HipMember, Mailbox, MemberShape};
use crate::status::{check, to_sumcheck_error};

/// One registered device member: raw handle, the mailbox its `prove_round` drains, and how many sums a round returns.
struct Entry(*mut ffi::jolt_member, Mailbox, usize);
// SAFETY: entries are only touched from the proving thread (`prove_batch` is single-threaded, prover.rs:124-146); the bound exists
// because `ProofSession` state must be `Send`.
unsafe impl Send for Entry {}

/// Per-proof carry parked in the `ProofSession`: the device members of the batch being proved, in `prepare` (= batch) order.
/// Shared between the scheduler (minted by the stage front BEFORE `begin_batch` prepares the members, stage3.rs:81-92) and the
/// `prepare` calls that fill it.
#[derive(Clone, Default)]
pub struct HipBatchCarry {
    entries: Arc<Mutex<Vec<Entry>>>,
}
crate::status::zero_host_heap!(HipBatchCarry);

pub(crate) fn register_member(session: &mut ProofSession, member: &HipMember) {
    let n_evals = match member.shape {
        MemberShape::Evals { degree } => degree + 1,
        MemberShape::SkippedOne { degree } => degree,
        MemberShape::GruenProduct => 2,
    };
    let carry = session.state_or_insert_with(HipBatchCarry::default);
    if let Ok(mut entries) = carry.entries.lock() {
        entries.push(Entry(member.raw, Rc::clone(&member.mailbox), n_evals));
    }
}

/// Factory stored in `JoltBackend::round_scheduler`.
pub struct HipBuildRoundScheduler {
    pub ctx: Arc<HipContext>,
}

impl BuildRoundScheduler<Fr> for HipBuildRoundScheduler {
    fn build(&self, session: &mut ProofSession) -> Box<dyn RoundScheduler<Fr>> {
        // a fresh carry per stage: the previous stage's members are gone
        let carry = HipBatchCarry::default();
        session.park(carry.clone());
        Box::new(HipRoundScheduler { ctx: Arc::clone(&self.ctx), carry })
    }
}

pub struct HipRoundScheduler {
    ctx: Arc<HipContext>,
    carry: HipBatchCarry,
}

impl RoundScheduler<Fr> for HipRoundScheduler {
    fn batch_prove_round(&mut self, work: &mut [MemberRound<'_, Fr>]) -> Result<(), SumcheckError<Fr>> {
        let _span = tracing::trace_span!("HipRoundScheduler::batch_prove_round", members = work.len()).entered();
        let entries = self.carry.entries.lock().map_err(|_| SumcheckError::MissingEvaluationSource { kind: "device" })?;
        // a batch whose members are not all device members (a slot served by another backend) degrades to per-member launches
        if !work.is_empty() && work.iter().all(|w| w.index < entries.len()) {
            let handles: Vec<*mut ffi::jolt_member> = work.iter().map(|w| entries[w.index].0).collect();
            let binds: Vec<*const ffi::jolt_fr_t> = work.iter().map(|w| w.bind.as_ref().map_or(ptr::null(), |b| (b as *const Fr).cast())).collect();
            let total: usize = work.iter().map(|w| entries[w.index].2).sum();
            let mut evals = vec![Fr::default(); total];
            // SAFETY: handle / bind arrays have one entry per member; `evals` has the concatenated capacity the header prescribes.
            check(unsafe { ffi::jolt_round_group_prove(self.ctx.raw, handles.as_ptr(), handles.len(), binds.as_ptr(), evals.as_mut_ptr().cast(), total) }, self.ctx.raw)
                .map_err(to_sumcheck_error)?;
            let mut off = 0;
            for w in work.iter() {
                let Entry(_, mailbox, n) = &entries[w.index];
                *mailbox.borrow_mut() = Some(evals[off..off + n].to_vec());
                off += n;
            }
        }
        drop(entries);
        // every member assembles its own message (round check, interpolation, gruen cubic): host work, O(degree)
        for item in work.iter_mut() {
            item.run()?;
        }
        Ok(())
    }

    fn batch_finish_rounds(&mut self, finishes: &mut [MemberFinish<'_, Fr>]) -> Result<(), SumcheckError<Fr>> {
        // the final binds are one launch per table group inside jolt_member_finish already; declaration order is fine
        for item in finishes.iter_mut() {
            item.run()?;
        }
        Ok(())
    }
}
