//! `StreamingCommitment` for [`HipHyperKzg`] (`crates/jolt-openings/src/schemes.rs:167-288`), plus the transparent-mode shims of
//! `ZkOpeningScheme` / `ZkStreamingCommitment` (`schemes.rs:301-365`).
//!
//! Why it is needed: both backend constructors bound their commit slot by `PCS: ModeStreamingCommitment`
//! (`crates/jolt-kernels/src/reference/mod.rs:88-92`, `optimized/mod.rs:136-139`; the alias is `StreamingCommitment` without the `zk`
//! feature and `ZkStreamingCommitment` with it, `crates/jolt-kernels/src/commitment.rs:34-45`), so
//! `JoltBackend::<Fr, HipHyperKzg>::optimized()` in [`crate::backend::mi355x`] only type-checks with this impl in scope.  It is also what
//! the FALLBACK commit slot runs when [`crate::backend::HipCommitWitness`] hands a grid back (`optimized/commitment.rs:73-115`).
//!
//! What streaming means for a KZG-type scheme.  A committed column is delivered in coefficient order of the shared commitment grid
//! (cycle-major: coefficient (address k, cycle j) at index `k * T + j`, dense columns at `k = 0`; `commitment.rs:86-130`):
//!
//! * dense columns as consecutive `row_width` windows -- `feed` / `feed_u64` / `feed_i128` / `feed_i128_rows_with` / `feed_zeros`
//!   (`reference/commitment.rs:86-121`, `optimized/commitment.rs:317-352`).  `kzg_commit` is linear in the coefficients
//!   (`crates/jolt-hyperkzg/src/kzg.rs:15-27`), so the partial commitment is the pair (running point, next coefficient index) and a window adds
//!   `sum_i values[i] * g1_powers[next + i]`.  Windows are STAGED on the host and flushed to the device as one run of up to 2^22 coefficients
//!   (`jolt_msm_g1_window`: upload, promotion of machine integers on the device, one MSM against the resident bases at the run's offset) --
//!   a 2^13-term MSM per window would be launch-latency on a GPU, and the running point is a plain value, so `PartialCommitment: Clone` holds;
//! * one-hot columns as `row_width` chunks of hot addresses -- `process_one_hot_chunk(s_with)` then `finish_one_hot_column_major_chunks`
//!   (`optimized/commitment.rs:338-352`).  The base a hot cycle selects is `g1_powers[k * T + j]` and `T` (= chunks x row_width) is only known
//!   at the finish, so a chunk "commitment" is the chunk's hot addresses (2 B per cycle, shared by `Arc`) and the finish uploads the column
//!   once and sums the selected bases on the device (`jolt_grid_commit_onehot`: additions only, no scalars).
//!
//! Failure policy: the trait's feed / finish methods return no `Result` (Dory asserts, `crates/jolt-dory/src/streaming.rs:52-63`).  Shape
//! violations assert like Dory's; a DEVICE failure (no HBM left, the device lost) is logged and the window falls back to the reference's own
//! `JoltGroup::msm` over the host bases -- the commitment is the same group element either way.
use std::sync::Arc;

use jolt_crypto::{Bn254G1, JoltGroup};
use jolt_field::{Fr, Ring};
use jolt_hyperkzg::HyperKZGCommitment;
use jolt_openings::{CommitmentScheme, OpeningsError, StreamingCommitment, ZkOpeningScheme, ZkStreamingCommitment};
use jolt_poly::MultilinearPoly;
use jolt_transcript::Transcript;

use crate::ffi;
use crate::ops::HipHotIndices;
use crate::pcs::{HipHyperKzg, HipHyperKzgSetup};
use crate::status::{check, HipError};

/// Coefficients staged before a flush: 2^22 field elements = 128 MiB of host memory at most, one large MSM instead of 512 short ones.
const FLUSH_AT: usize = 1 << 22;

/// The windows fed since the last flush, in the form they arrived in (promotion to `Fr` happens on the device).
#[derive(Clone, Debug)]
enum Staged {
    Empty,
    Field(Vec<Fr>),
    U64(Vec<u64>),
    I128(Vec<i128>),
}

impl Staged {
    fn len(&self) -> usize {
        match self {
            Self::Empty => 0,
            Self::Field(v) => v.len(),
            Self::U64(v) => v.len(),
            Self::I128(v) => v.len(),
        }
    }
}

/// `StreamingCommitment::PartialCommitment`: the commitment of the coefficients flushed so far, the index of the next coefficient, and
/// the staged windows `[next - staged.len(), next)`.
#[derive(Clone, Debug)]
pub struct HipPartialCommitment {
    point: Bn254G1,
    next: usize,
    staged: Staged,
}

impl HipPartialCommitment {
    /// `point += sum_i staged[i] * g1_powers[base + i]` on the device; on a device failure the same sum from the host bases.
    fn flush(&mut self, setup: &HipHyperKzgSetup) {
        let staged = std::mem::replace(&mut self.staged, Staged::Empty);
        let n = staged.len();
        if n == 0 {
            return;
        }
        let base = self.next - n;
        let (kind, host): (i32, *const core::ffi::c_void) = match &staged {
            Staged::Empty => return,
            Staged::Field(v) => (ffi::JOLT_SCALAR_FR, v.as_ptr().cast()),
            Staged::U64(v) => (ffi::JOLT_INT_U64, v.as_ptr().cast()),
            // i128 is two little-endian u64 words, low first, two's complement: the layout JOLT_INT_I128 names
            Staged::I128(v) => (ffi::JOLT_INT_I128, v.as_ptr().cast()),
        };
        let acc = self.point;
        let device = setup.with_device(|ctx, srs| {
            let mut out = Bn254G1::default();
            // SAFETY: live handles of one context; `host` points at `n` values of the stated kind that outlive the (synchronous) call;
            // `acc` / `out` are one jolt_g1_t each (Bn254G1 is repr(transparent) over ark's projective point, crates/jolt-crypto/src/ec/bn254/mod.rs:17-24).
            check(
                unsafe { ffi::jolt_msm_g1_window(ctx.raw, srs.raw, base, kind, host, n, (&acc as *const Bn254G1).cast(), (&mut out as *mut Bn254G1).cast()) },
                ctx.raw,
            )?;
            Ok(out)
        });
        self.point = match device {
            Ok(point) => point,
            Err(e) => acc + host_window(setup, base, &staged, &e),
        };
    }

    fn make_room(&mut self, incoming: usize, same_kind: bool, setup: &HipHyperKzgSetup) {
        if !same_kind || self.staged.len() + incoming > FLUSH_AT {
            self.flush(setup);
        }
    }
}

/// The reference's own arithmetic for one run (`kzg_commit`'s `JoltGroup::msm` over the prefix slice, `kzg.rs:19-26`), used only when the device call failed.
fn host_window(setup: &HipHyperKzgSetup, base: usize, staged: &Staged, why: &HipError) -> Bn254G1 {
    tracing::error!(status = why.status, detail = %why.detail, "jolt_msm_g1_window failed; committing this run on the host");
    let bases = setup.inner.g1_powers();
    let n = staged.len();
    assert!(base + n <= bases.len(), "streaming: coefficients [{base}, {}) exceed the HyperKZG SRS ({} powers)", base + n, bases.len());
    let scalars: Vec<Fr> = match staged {
        Staged::Empty => Vec::new(),
        Staged::Field(v) => v.clone(),
        Staged::U64(v) => v.iter().copied().map(<Fr as Ring>::from_u64).collect(),
        Staged::I128(v) => v.iter().copied().map(<Fr as Ring>::from_i128).collect(),
    };
    <Bn254G1 as JoltGroup>::msm(&bases[base..base + n], &scalars)
}

/// `StreamingCommitment::OneHotStreamContext`: nothing device-side is needed before the finish (Dory caches affine bases here,
/// `crates/jolt-dory/src/streaming.rs:210-228`; ours are resident already); the row width is kept to check the chunks against.
#[derive(Clone, Copy, Debug)]
pub struct HipOneHotStream {
    row_width: usize,
}

/// `StreamingCommitment::OneHotChunkCommitment`: the chunk's hot addresses (`u16::MAX` = cold cycle).  The group work happens in
/// `finish_one_hot_column_major_chunks`, where the cycle count -- and with it the base `k * T + j` of every hot cycle -- is known.
#[derive(Clone, Debug)]
pub struct HipOneHotChunk {
    hot: Arc<[u16]>,
}

const COLD: u16 = u16::MAX;

fn encode_chunk(one_hot_k: usize, addresses: impl Iterator<Item = Option<usize>>) -> HipOneHotChunk {
    assert!(one_hot_k != 0 && one_hot_k < usize::from(COLD), "streaming one-hot: one_hot_k ({one_hot_k}) must be in 1..65535");
    let hot: Vec<u16> = addresses
        .map(|a| match a {
            None => COLD,
            Some(k) => {
                assert!(k < one_hot_k, "streaming one-hot: hot row {k} outside k={one_hot_k}");
                // k < one_hot_k < 65535: the conversion cannot truncate
                u16::try_from(k).unwrap_or(COLD)
            }
        })
        .collect();
    HipOneHotChunk { hot: hot.into() }
}

impl StreamingCommitment for HipHyperKzg {
    type PartialCommitment = HipPartialCommitment;
    type OneHotChunkCommitment = HipOneHotChunk;
    type OneHotStreamContext = HipOneHotStream;

    fn begin(_setup: &Self::ProverSetup) -> Self::PartialCommitment {
        HipPartialCommitment { point: Bn254G1::default(), next: 0, staged: Staged::Empty }
    }

    fn feed(partial: &mut Self::PartialCommitment, chunk: &[Fr], setup: &Self::ProverSetup) {
        let same_kind = matches!(partial.staged, Staged::Field(_) | Staged::Empty);
        partial.make_room(chunk.len(), same_kind, setup);
        match &mut partial.staged {
            Staged::Field(v) => v.extend_from_slice(chunk),
            staged => *staged = Staged::Field(chunk.to_vec()),
        }
        partial.next += chunk.len();
    }

    fn finish(mut partial: Self::PartialCommitment, setup: &Self::ProverSetup) -> Self::Output {
        partial.flush(setup);
        HyperKZGCommitment { point: partial.point }
    }

    /// Zero coefficients add nothing: the staged run is closed and the offset moves on.
    fn feed_zeros(partial: &mut Self::PartialCommitment, row_width: usize, rows: usize, setup: &Self::ProverSetup) {
        if rows == 0 {
            return;
        }
        partial.flush(setup);
        partial.next += row_width * rows;
    }

    fn feed_u64(partial: &mut Self::PartialCommitment, chunk: &[u64], setup: &Self::ProverSetup) {
        let same_kind = matches!(partial.staged, Staged::U64(_) | Staged::Empty);
        partial.make_room(chunk.len(), same_kind, setup);
        match &mut partial.staged {
            Staged::U64(v) => v.extend_from_slice(chunk),
            staged => *staged = Staged::U64(chunk.to_vec()),
        }
        partial.next += chunk.len();
    }

    fn feed_i128(partial: &mut Self::PartialCommitment, chunk: &[i128], setup: &Self::ProverSetup) {
        let same_kind = matches!(partial.staged, Staged::I128(_) | Staged::Empty);
        partial.make_room(chunk.len(), same_kind, setup);
        match &mut partial.staged {
            Staged::I128(v) => v.extend_from_slice(chunk),
            staged => *staged = Staged::I128(chunk.to_vec()),
        }
        partial.next += chunk.len();
    }

    /// The batch entry point of the optimized commit kernel (`optimized/commitment.rs:330-337`): the whole superchunk is ONE staged run --
    /// the row structure only matters to schemes that commit row by row.
    fn feed_i128_rows_with(partial: &mut Self::PartialCommitment, value: impl Fn(usize) -> i128 + Sync, count: usize, row_width: usize, setup: &Self::ProverSetup) {
        assert!(row_width != 0 && count % row_width == 0, "streaming: batch length ({count}) must be a multiple of the row width ({row_width})");
        let same_kind = matches!(partial.staged, Staged::I128(_) | Staged::Empty);
        partial.make_room(count, same_kind, setup);
        match &mut partial.staged {
            Staged::I128(v) => v.extend((0..count).map(&value)),
            staged => *staged = Staged::I128((0..count).map(&value).collect()),
        }
        partial.next += count;
    }

    fn begin_one_hot_column_major_stream(_setup: &Self::ProverSetup, row_width: usize) -> Self::OneHotStreamContext {
        assert!(row_width.is_power_of_two(), "streaming one-hot: row width ({row_width}) must be a power of two");
        HipOneHotStream { row_width }
    }

    fn process_one_hot_chunk(context: &mut Self::OneHotStreamContext, _setup: &Self::ProverSetup, one_hot_k: usize, chunk: &[Option<usize>]) -> Self::OneHotChunkCommitment {
        assert!(chunk.len() <= context.row_width, "streaming one-hot: chunk length ({}) exceeds the row width ({})", chunk.len(), context.row_width);
        encode_chunk(one_hot_k, chunk.iter().copied())
    }

    fn process_one_hot_chunks_with(
        context: &mut Self::OneHotStreamContext,
        _setup: &Self::ProverSetup,
        one_hot_k: usize,
        hot_address: impl Fn(usize) -> Option<usize> + Sync,
        count: usize,
        chunk_width: usize,
    ) -> Vec<Self::OneHotChunkCommitment> {
        assert!(chunk_width != 0 && chunk_width <= context.row_width, "streaming one-hot: chunk width ({chunk_width}) must be in 1..={}", context.row_width);
        (0..count).step_by(chunk_width).map(|base| encode_chunk(one_hot_k, (base..(base + chunk_width).min(count)).map(&hot_address))).collect()
    }

    fn finish_one_hot_column_major_chunks(setup: &Self::ProverSetup, one_hot_k: usize, chunks: &[Self::OneHotChunkCommitment]) -> (Self::Output, Self::OpeningHint) {
        assert!(one_hot_k != 0, "streaming one-hot: one_hot_k must be nonzero");
        assert!(!chunks.is_empty(), "streaming one-hot: cannot finish an empty chunk list");
        let cycles: usize = chunks.iter().map(|c| c.hot.len()).sum();
        assert!(cycles.is_power_of_two(), "streaming one-hot: the chunks hold {cycles} cycles, not a power of two");
        let point = match one_hot_grid_commit(setup, one_hot_k, cycles, chunks) {
            Ok(point) => point,
            Err(e) => {
                tracing::error!(status = e.status, detail = %e.detail, "jolt_grid_commit_onehot failed; committing this column on the host");
                let bases = setup.inner.g1_powers();
                assert!(one_hot_k * cycles <= bases.len(), "streaming one-hot: a {one_hot_k} x {cycles} grid exceeds the HyperKZG SRS ({} powers)", bases.len());
                let mut sum = Bn254G1::default();
                for (j, &hot) in chunks.iter().flat_map(|c| c.hot.iter()).enumerate() {
                    if hot != COLD {
                        sum += bases[usize::from(hot) * cycles + j];
                    }
                }
                sum
            }
        };
        (HyperKZGCommitment { point }, Self::OpeningHint::default())
    }
}

/// The column uploaded once (1 B per cycle up to k = 255, 2 B beyond) and the bases it selects summed on the device.
fn one_hot_grid_commit(setup: &HipHyperKzgSetup, one_hot_k: usize, cycles: usize, chunks: &[HipOneHotChunk]) -> Result<Bn254G1, HipError> {
    let k = u32::try_from(one_hot_k).map_err(|_| HipError::size_mismatch("one_hot_k beyond u32"))?;
    setup.with_device(|ctx, srs| {
        let column = if one_hot_k <= 255 {
            let narrow: Vec<u8> = chunks.iter().flat_map(|c| c.hot.iter()).map(|&h| if h == COLD { 0xFF } else { u8::try_from(h).unwrap_or(0xFF) }).collect();
            HipHotIndices::upload(ctx, &narrow, 1, cycles, k)?
        } else {
            let wide: Vec<u16> = chunks.iter().flat_map(|c| c.hot.iter().copied()).collect();
            HipHotIndices::upload16(ctx, &wide, 1, cycles, k)?
        };
        column.grid_commit(srs)?.into_iter().next().ok_or_else(|| HipError::size_mismatch("one commitment per one-hot column"))
    })
}

/// HyperKZG has no hiding mode.  `jolt_prover::dory::prove` nevertheless bounds its PCS by
/// `ZkOpeningScheme<HidingCommitment = VC::Output, Blind = F>` in BOTH compiled modes (`crates/jolt-prover/src/dory/prover.rs:120-130`), and
/// only CALLS the hiding methods when the proof mode is ZK (`crates/jolt-openings/src/schemes.rs:682`, behind `ZkBatchOpeningScheme`).  The shim
/// therefore satisfies the bound for the transparent mode -- `HidingCommitment` is the Pedersen vector commitment's output the prover's `VC`
/// parameter names (`crates/jolt-crypto/src/ec/pedersen.rs:64-66`: `Pedersen<Bn254G1>::Output = Bn254G1`), `Blind` the field -- and every hiding method answers
/// `OpeningsError::InvalidBatch`: a ZK proof over this scheme is refused, never silently transparent.
impl ZkOpeningScheme for HipHyperKzg {
    type HidingCommitment = Bn254G1;
    type Blind = Fr;

    fn commit_zk<P: MultilinearPoly<Self::Field> + ?Sized>(_poly: &P, _setup: &Self::ProverSetup) -> Result<(Self::Output, Self::OpeningHint), OpeningsError> {
        Err(no_hiding_mode())
    }

    fn open_zk<P: MultilinearPoly<Self::Field> + ?Sized>(
        _poly: &P,
        _point: &[Self::Field],
        _eval: Self::Field,
        _setup: &Self::ProverSetup,
        _hint: Self::OpeningHint,
        _transcript: &mut impl Transcript<Challenge = Self::Field>,
    ) -> Result<(Self::Proof, Self::HidingCommitment, Self::Blind), OpeningsError> {
        Err(no_hiding_mode())
    }

    fn verify_zk(
        _commitment: &Self::Output,
        _point: &[Self::Field],
        _proof: &Self::Proof,
        _setup: &Self::VerifierSetup,
        _transcript: &mut impl Transcript<Challenge = Self::Field>,
    ) -> Result<Self::HidingCommitment, OpeningsError> {
        Err(no_hiding_mode())
    }
}

fn no_hiding_mode() -> OpeningsError {
    OpeningsError::InvalidBatch("HyperKZG has no hiding mode: build jolt-kernels without the `zk` feature for the MI355X backend".to_owned())
}

/// With the `zk` feature `ModeStreamingCommitment` is `ZkStreamingCommitment` (`crates/jolt-kernels/src/commitment.rs:42-45`): the bound is met so the
/// crate compiles in either mode.  The hiding finishes have no transparent meaning: the trait gives them no `Result`, so they REFUSE by panicking -- a
/// non-hiding commitment must never be produced (and absorbed, and sent) under a ZK proof; `open_zk` above refuses the same way one step later for callers that
/// committed elsewhere.  (HyperKZG in this crate is the transparent scheme; a hiding variant is out of the hot path's scope)
impl ZkStreamingCommitment for HipHyperKzg {
    fn finish_zk_with_hint(_partial: Self::PartialCommitment, _setup: &Self::ProverSetup) -> (Self::Output, Self::OpeningHint) {
        tracing::error!("HipHyperKzg has no hiding commitment: a ZK proof over this scheme is refused at its first commitment");
        panic!("HipHyperKzg::finish_zk_with_hint: the device HyperKZG is transparent; refusing to return a non-hiding commitment under the zk feature")
    }

    fn finish_zk_one_hot_column_major_chunks(_setup: &Self::ProverSetup, _one_hot_k: usize, _chunks: &[Self::OneHotChunkCommitment]) -> (Self::Output, Self::OpeningHint) {
        tracing::error!("HipHyperKzg has no hiding commitment: a ZK proof over this scheme is refused at its first commitment");
        panic!("HipHyperKzg::finish_zk_one_hot_column_major_chunks: the device HyperKZG is transparent; refusing to return a non-hiding commitment under the zk feature")
    }
}
