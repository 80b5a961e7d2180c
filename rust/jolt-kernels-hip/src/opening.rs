//! Stage 8 on the device: the opening hint the commit slot leaves behind, the joint-opening slot, and the batch opening that consumes both.
//!
//! Reference seams:
//!   * `CommitmentScheme::OpeningHint` (`crates/jolt-openings/src/schemes.rs:49-50`, "auxiliary data from commit reused during opening") -> [`HipOpeningHint`]:
//!     the committed column as it lies in HBM (a column of a resident one-hot block, or a dense table), and -- through the block -- the class sums of
//!     `jolt_grid_hint_begin`, enqueued at commit time in the background and consumed by the opening's first level commitments;
//!   * `AdditivelyHomomorphic::combine_hints(hints, scalars)` (`schemes.rs:157-162`), which `HomomorphicBatch::prove_batch` calls with the batch's RLC scalars right
//!     before `PCS::open(&joint, .., Some(combined_hint), ..)` (`schemes.rs:508-521`) -> [`HipOpeningHint::Joint`]: everything `open` needs to build the joint
//!     polynomial ON the device (`jolt_grid_joint_polynomial`) and open it with `jolt_host_hyperkzg_open_grid` -- the host-side `RlcSource` is then never densified;
//!   * `JointOpeningPolynomials` (`crates/jolt-kernels/src/opening.rs:42-54`) -> [`HipJointOpening`]: the committed polynomials as [`HipGridColumn`]s, grid-embedded
//!     `MultilinearPoly` objects whose `evaluate` runs on the device and whose row interface materialises on the host only if a generic consumer asks for it.
//!
//! Written blind (no Rust toolchain in this image).
use std::borrow::Cow;
use std::collections::BTreeMap;
use std::ptr;
use std::sync::{Arc, Mutex, OnceLock};

use jolt_claims::protocols::jolt::{JoltCommittedPolynomial, TracePolynomialOrder};
use jolt_field::{Fr, Ring};
use jolt_kernels::commitment::CommitmentGrid;
use jolt_kernels::opening::JointOpeningPolynomials;
use jolt_kernels::{KernelError, ProofSession};
use jolt_poly::MultilinearPoly;
use jolt_witness::JoltWitnessPlane;

use crate::context::{HipContext, HipTable};
use crate::ffi;
use crate::msm::HipSrs;
use crate::ops::HipHotIndices;
use crate::status::{check, HipError};

/// An owned `jolt_grid_hint`: the class sums of a block's columns for the first `levels` level commitments, in flight from the moment it is created.
pub struct HipGridHint {
    ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_grid_hint,
    pub levels: u32,
}
// SAFETY: see HipContext.
unsafe impl Send for HipGridHint {}
unsafe impl Sync for HipGridHint {}

impl HipGridHint {
    /// `jolt_grid_hint_begin` over ONE block (the columns of the hint are the block's columns, in order).
    pub fn begin(ctx: &Arc<HipContext>, srs: &HipSrs, block: &HipHotIndices, levels: u32, background: bool) -> Result<Self, HipError> {
        let sources = [block.raw.cast_const()];
        let mut raw = ptr::null_mut();
        // SAFETY: live SRS and block; one source; valid out-pointer.
        check(unsafe { ffi::jolt_grid_hint_begin(ctx.raw, srs.raw, sources.as_ptr(), 1, levels, i32::from(background), &mut raw) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw, levels })
    }
}
impl Drop for HipGridHint {
    fn drop(&mut self) {
        // SAFETY: owned handle, freed once (the library orders the free behind the sums).
        let _ = unsafe { ffi::jolt_grid_hint_free(self.ctx.raw, self.raw) };
    }
}

/// The one-hot columns of a commitment, resident as ONE block (`n_columns x T` hot indices, K = 2^log_k_chunk), with the class sums begun at commit time.
pub struct ResidentGridBlock {
    pub ctx: Arc<HipContext>,
    pub indices: HipHotIndices,
    pub n_columns: usize,
    pub log_t: usize,
    pub log_k: usize,
    /// taken by the opening that consumes it
    pub hint: Mutex<Option<HipGridHint>>,
    host: OnceLock<Vec<u8>>,
}
// SAFETY: the device handles are used from one thread at a time (see HipContext); the host cache is write-once.
unsafe impl Sync for ResidentGridBlock {}
unsafe impl Send for ResidentGridBlock {}

impl ResidentGridBlock {
    pub fn new(ctx: &Arc<HipContext>, indices: HipHotIndices, n_columns: usize, log_t: usize, log_k: usize, hint: Option<HipGridHint>) -> Self {
        Self { ctx: Arc::clone(ctx), indices, n_columns, log_t, log_k, hint: Mutex::new(hint), host: OnceLock::new() }
    }
    fn host_indices(&self) -> &[u8] {
        self.host.get_or_init(|| {
            let mut out = vec![0xFFu8; self.n_columns << self.log_t];
            // SAFETY: `out` holds n_columns x T bytes, the size the library writes.
            let _ = unsafe { ffi::jolt_onehot_download(self.ctx.raw, self.indices.raw, out.as_mut_ptr()) };
            out
        })
    }
}

/// `CommitmentScheme::OpeningHint` of [`crate::pcs::HipHyperKzg`].
#[derive(Clone, Default)]
pub enum HipOpeningHint {
    /// a commitment made through the generic `commit` (no resident column): the opening uploads the polynomial
    #[default]
    None,
    OneHot { block: Arc<ResidentGridBlock>, column: usize },
    /// a dense column of T entries at address 0 of the grid
    Dense { table: Arc<HipTable>, log_k: usize },
    /// `combine_hints`: the members of a homomorphic batch with their RLC scalars
    Joint(Arc<JointHint>),
}

pub struct JointHint {
    pub block: Option<Arc<ResidentGridBlock>>,
    /// one scalar per column of `block` (zero for a column the batch does not open)
    pub onehot_scalars: Vec<Fr>,
    pub dense: Vec<(Arc<HipTable>, Fr)>,
    pub log_k: usize,
}
// SAFETY: as ResidentGridBlock.
unsafe impl Sync for JointHint {}
unsafe impl Send for JointHint {}

/// `AdditivelyHomomorphic::combine_hints`: `None` as soon as one member has no resident column (the opening then takes the upload path) or the one-hot members come
/// from different blocks.
pub fn combine_hints(hints: Vec<HipOpeningHint>, scalars: &[Fr]) -> HipOpeningHint {
    let mut block: Option<Arc<ResidentGridBlock>> = None;
    let mut onehot_scalars: Vec<Fr> = Vec::new();
    let mut dense = Vec::new();
    let mut log_k = 0usize;
    for (hint, scalar) in hints.into_iter().zip(scalars) {
        match hint {
            HipOpeningHint::OneHot { block: b, column } => {
                if let Some(have) = &block {
                    if !Arc::ptr_eq(have, &b) {
                        return HipOpeningHint::None;
                    }
                } else {
                    onehot_scalars = vec![Fr::default(); b.n_columns];
                    log_k = b.log_k;
                    block = Some(b);
                }
                if column >= onehot_scalars.len() {
                    return HipOpeningHint::None;
                }
                onehot_scalars[column] += *scalar;
            }
            HipOpeningHint::Dense { table, log_k: k } => {
                if log_k == 0 {
                    log_k = k;
                }
                dense.push((table, *scalar));
            }
            HipOpeningHint::None | HipOpeningHint::Joint(_) => return HipOpeningHint::None,
        }
    }
    HipOpeningHint::Joint(Arc::new(JointHint { block, onehot_scalars, dense, log_k }))
}

impl JointHint {
    /// The joint polynomial of the batch, built on the device in one pass (`jolt_grid_joint_polynomial`; `RlcSource::to_dense` of `schemes.rs:559-573` never runs).
    pub fn joint_polynomial(&self, ctx: &Arc<HipContext>) -> Result<HipTable, HipError> {
        let sources: Vec<*const ffi::jolt_onehot> = self.block.iter().map(|b| b.indices.raw.cast_const()).collect();
        let dense: Vec<*mut ffi::jolt_table> = self.dense.iter().map(|(t, _)| t.raw).collect();
        let dense_scalars: Vec<Fr> = self.dense.iter().map(|(_, s)| *s).collect();
        let mut raw = ptr::null_mut();
        // SAFETY: live handles; one scalar per one-hot column / dense table; a NULL array goes with a zero count.
        check(
            unsafe {
                ffi::jolt_grid_joint_polynomial(
                    ctx.raw,
                    if sources.is_empty() { ptr::null() } else { sources.as_ptr() },
                    sources.len(),
                    if sources.is_empty() { ptr::null() } else { self.onehot_scalars.as_ptr().cast() },
                    if dense.is_empty() { ptr::null() } else { dense.as_ptr() },
                    dense.len(),
                    if dense.is_empty() { ptr::null() } else { dense_scalars.as_ptr().cast() },
                    self.log_k as u32,
                    &mut raw,
                )
            },
            ctx.raw,
        )?;
        Ok(HipTable { ctx: Arc::clone(ctx), raw })
    }
}

/// One committed polynomial embedded over the commitment grid (cycle-major: index = address * T + cycle), as the batch opening sees it.
pub struct HipGridColumn {
    hint: HipOpeningHint,
    grid_vars: usize,
    log_t: usize,
    dense_host: OnceLock<Vec<Fr>>,
}
// SAFETY: as ResidentGridBlock.
unsafe impl Sync for HipGridColumn {}
unsafe impl Send for HipGridColumn {}

impl HipGridColumn {
    pub fn hint(&self) -> &HipOpeningHint {
        &self.hint
    }
    fn host(&self) -> &[Fr] {
        self.dense_host.get_or_init(|| {
            let mut out = vec![Fr::default(); 1usize << self.grid_vars];
            let cycles = 1usize << self.log_t;
            match &self.hint {
                HipOpeningHint::OneHot { block, column } => {
                    let hot = &block.host_indices()[column * cycles..(column + 1) * cycles];
                    for (j, k) in hot.iter().enumerate() {
                        if *k != 0xFF {
                            out[usize::from(*k) * cycles + j] = Fr::from_u64(1);
                        }
                    }
                }
                HipOpeningHint::Dense { table, .. } => {
                    if let Ok(values) = table.download() {
                        out[..values.len().min(cycles)].copy_from_slice(&values[..values.len().min(cycles)]);
                    }
                }
                HipOpeningHint::None | HipOpeningHint::Joint(_) => {}
            }
            out
        })
    }
}

impl MultilinearPoly<Fr> for HipGridColumn {
    fn num_vars(&self) -> usize {
        self.grid_vars
    }
    /// On the device: a one-hot column at `[r_address || r_cycle]` is `sum_j eq(r_cycle, j) eq(r_address, hot(j))` (fold the address axis, evaluate the cycles); a dense
    /// column sits at address 0: `eq(r_address, 0) * f(r_cycle)`.
    fn evaluate(&self, point: &[Fr]) -> Fr {
        let log_k = self.grid_vars - self.log_t;
        let (r_address, r_cycle) = point.split_at(log_k.min(point.len()));
        let on_device = || -> Result<Fr, HipError> {
            match &self.hint {
                HipOpeningHint::OneHot { block, column } => {
                    let scale = block.ctx.eq_evals(r_address, None)?;
                    let mut raw = ptr::null_mut();
                    // SAFETY: live block and scale table (K entries); valid out-pointer.
                    check(unsafe { ffi::jolt_onehot_materialize(block.ctx.raw, block.indices.raw, *column, scale.raw, &mut raw) }, block.ctx.raw)?;
                    HipTable { ctx: Arc::clone(&block.ctx), raw }.evaluate(r_cycle)
                }
                HipOpeningHint::Dense { table, .. } => {
                    let zero_weight = r_address.iter().fold(Fr::from_u64(1), |acc, r| acc * (Fr::from_u64(1) - *r));
                    Ok(zero_weight * table.evaluate(r_cycle)?)
                }
                HipOpeningHint::None | HipOpeningHint::Joint(_) => Err(HipError::size_mismatch("no resident column")),
            }
        };
        on_device().unwrap_or_else(|_| jolt_poly::Polynomial::new(self.host().to_vec()).evaluate(point))
    }
    fn for_each_row(&self, sigma: usize, f: &mut dyn FnMut(usize, &[Fr])) {
        for (i, row) in self.host().chunks(1usize << sigma).enumerate() {
            f(i, row);
        }
    }
    fn dense_evaluations(&self) -> Option<&[Fr]> {
        None // lazily materialised only through for_each_row / to_dense
    }
    fn to_dense(&self) -> Cow<'_, [Fr]> {
        Cow::Borrowed(self.host())
    }
    fn is_one_hot(&self) -> bool {
        matches!(self.hint, HipOpeningHint::OneHot { .. })
    }
    fn one_hot_k(&self) -> Option<usize> {
        match &self.hint {
            HipOpeningHint::OneHot { block, .. } => Some(1usize << block.log_k),
            _ => None,
        }
    }
    fn for_each_one(&self, f: &mut dyn FnMut(usize)) {
        if let HipOpeningHint::OneHot { block, column } = &self.hint {
            let cycles = 1usize << self.log_t;
            for (j, k) in block.host_indices()[column * cycles..(column + 1) * cycles].iter().enumerate() {
                if *k != 0xFF {
                    f(usize::from(*k) * cycles + j);
                }
            }
        }
    }
}

/// What the commit slot parks for stage 8: the resident column behind every committed polynomial it served.
#[derive(Default)]
pub struct ResidentCommitted(pub BTreeMap<JoltCommittedPolynomial, HipOpeningHint>);
crate::status::zero_host_heap!(ResidentCommitted);

/// `backend.joint_opening`: the committed polynomials in final-opening batch order, embedded over the grid -- from the columns the commit slot left resident
/// ([`ResidentCommitted`]); anything else (advice, precommitted program tables, address-major order) goes to the fallback slot, as `HipCommitWitness` does.
pub struct HipJointOpening {
    pub ctx: Arc<HipContext>,
    pub fallback: Box<dyn JointOpeningPolynomials<Fr>>,
}

impl JointOpeningPolynomials<Fr> for HipJointOpening {
    #[tracing::instrument(skip_all, name = "HipJointOpening::prepare")]
    fn prepare(
        &self,
        session: &mut ProofSession,
        witness: &dyn JoltWitnessPlane<Fr>,
        polynomials: &[JoltCommittedPolynomial],
        precommitted_tables: &BTreeMap<JoltCommittedPolynomial, Vec<Fr>>,
        grid: CommitmentGrid,
    ) -> Result<Vec<Box<dyn MultilinearPoly<Fr>>>, KernelError<Fr>> {
        let resident = session.state::<ResidentCommitted>();
        let served = grid.order == TracePolynomialOrder::CycleMajor
            && grid.total_vars == grid.log_k_chunk + grid.log_t
            && resident.is_some_and(|r| polynomials.iter().all(|id| r.0.contains_key(id)));
        if !served {
            return self.fallback.prepare(session, witness, polynomials, precommitted_tables, grid);
        }
        let resident = resident.ok_or(KernelError::InvariantViolation { reason: "the commit slot parked no resident columns" })?;
        Ok(polynomials
            .iter()
            .map(|id| {
                let hint = resident.0.get(id).cloned().unwrap_or_default();
                Box::new(HipGridColumn { hint, grid_vars: grid.total_vars, log_t: grid.log_t, dense_host: OnceLock::new() }) as Box<dyn MultilinearPoly<Fr>>
            })
            .collect())
    }
}
