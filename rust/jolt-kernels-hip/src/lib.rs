//! `jolt-kernels-hip`: the MI355X backend of the Jolt prover's sumcheck and polynomial-commitment hot path.
//!
//! The crate is a thin, safe layer over the C ABI of `libjolt_hip.so` (`include/jolt_hip.h`, raw declarations in [`ffi`]) that
//! implements the reference's OWN trait surface for this path, so `jolt-prover`'s stage drivers call it as a drop-in:
//!
//! | reference seam | here |
//! |---|---|
//! | `jolt_sumcheck::ProveRounds` (`crates/jolt-sumcheck/src/prover.rs:52-72`) | [`member::HipMember`] |
//! | `jolt_kernels::SumcheckKernel` (`crates/jolt-kernels/src/kernel.rs:72-126`) | [`member::HipSumcheckProver`] |
//! | `jolt_kernels::PrepareKernel` (`crates/jolt-kernels/src/backend.rs:98-111`) | [`member::HipPrepare`] (device twin of `NaiveSumcheckProver::new`) |
//! | `jolt_sumcheck::RoundScheduler` / `BuildRoundScheduler` (`prover.rs:110-120`, `backend.rs:68-70`) | [`scheduler::HipRoundScheduler`] |
//! | `JoltGroup::msm` for `Bn254G1` (`crates/jolt-crypto/src/ec/group.rs:63-70`) | [`msm::msm_g1`] |
//! | `optimized::{spartan_outer, spartan_product, ram_read_write, instruction_read_raf}` T-scale loops | [`ops::SpartanSums`], [`ops::HipRwMatrix`], [`ops::HipReadRaf`] |
//! | HyperKZG prover pieces (`crates/jolt-hyperkzg/src/{kzg,scheme}.rs`) | [`msm::HipSrs`], `ffi::jolt_hyperkzg_*` |
//! | `CommitmentScheme` + `AdditivelyHomomorphic` for HyperKZG (`crates/jolt-openings/src/schemes.rs:43-163`, `crates/jolt-hyperkzg/src/scheme.rs:275-353`) | [`pcs::HipHyperKzg`] over device-resident [`pcs::HipPoly`]s, the caller's transcript through `jolt_open_transcript_fn` |
//! | `StreamingCommitment` (+ transparent-mode `ZkOpeningScheme` / `ZkStreamingCommitment` shims) for HyperKZG (`crates/jolt-openings/src/schemes.rs:167-365`): what `JoltBackend::{reference,optimized}()` bound their commit slot by (`crates/jolt-kernels/src/commitment.rs:34-45`) | [`streaming`]: staged windows through `jolt_msm_g1_window`, one-hot columns through `jolt_grid_commit_onehot` |
//! | `RowSource::rows` / `WitnessBundle::from_row` witness hand-over (`crates/jolt-witness/src/consumer.rs:129-143`) | [`rows::HipPinnedRows`], [`rows::HipRows`]: one H2D copy of packed rows, columns extracted on the device |
//! | the stage-operator slots: `spartan_{outer,product}_remainder`, `ram_read_write`, `registers_read_write`, `instruction_read_raf`, `booleanity_address`, `bytecode_read_raf_{address,cycle}`, `hamming_weight_claim_reduction`, `ram_raf_evaluation`, `ram_output_check` (`crates/jolt-kernels/src/backend.rs:126-171`) | [`stage`]: one `PrepareKernel` per slot over a `jolt_stage_op` |
//! | the eleven cycle-domain relation slots of stages 2 - 6b (`crates/jolt-kernels/src/reference/*.rs`) | [`leaves`]: one `ResolveLeaves` per relation for [`member::HipPrepare`] |
//! | `CommitmentScheme::OpeningHint`, `AdditivelyHomomorphic::combine_hints`, `JointOpeningPolynomials` (`crates/jolt-openings/src/schemes.rs:49-50,157-162`, `crates/jolt-kernels/src/opening.rs:42-54`) | [`opening`]: resident columns + commit-time class sums -> `jolt_host_hyperkzg_open_grid` |
//! | `UniskipKernel`, `CommitWitness`, the backend constructor (`crates/jolt-kernels/src/{uniskip.rs:28-54, commitment.rs:137-160, optimized/mod.rs:136-196}`) | [`backend::HipUniskip`], [`backend::HipCommitWitness`], [`backend::mi355x`] |
//!
//! Host code stays Rust: Fiat-Shamir, claim wiring, round-polynomial assembly (`UnivariatePoly::from_evals`,
//! `gruen_poly_deg_3`) and error types are the reference's; only table-sized work crosses the boundary, and per round only
//! `Option<F>` goes down and `degree + 1` field elements come back (`specs/clean-slate-prover.md:565-573`).
#![deny(unsafe_op_in_unsafe_fn)]

pub mod ffi;
pub mod context;
pub mod leaves;
pub mod member;
pub mod msm;
pub mod opening;
pub mod ops;
pub mod pcs;
pub mod rows;
pub mod backend;
pub mod scheduler;
pub mod stage;
pub mod status;
pub mod streaming;

pub use context::{HipContext, HipTable};
pub use member::{HipMember, HipPrepare, HipSumcheckProver, MemberShape, MemberSlot};
pub use msm::{msm_cache_clear, msm_cache_evict, msm_g1, HipShardedOpening, HipSrs, SharedMsmContext};
pub use ops::{HipHotIndices, HipInts, HipKeyIndex, HipReadRaf, HipRegistersRw, HipRwMatrix, RwRow, SpartanSums};
pub use backend::{mi355x, with_relation, HipCommitWitness, HipUniskip, Mi355xParts, NodeWeights};
pub use pcs::{HipHyperKzg, HipHyperKzgSetup, HipPoly};
pub use rows::{HipPinnedRows, HipRows, HipRowsInFlight};
pub use scheduler::{HipBuildRoundScheduler, HipRoundScheduler};
pub use status::HipError;
pub use streaming::{HipOneHotChunk, HipOneHotStream, HipPartialCommitment};
pub use opening::{HipGridColumn, HipGridHint, HipJointOpening, HipOpeningHint, ResidentGridBlock};
pub use stage::{HipStageKernel, HipStageOp, ResidentTrace};
