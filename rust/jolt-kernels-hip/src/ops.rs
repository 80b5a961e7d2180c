//! Safe wrappers of the T-scale operators outside the generic member interface (`SURVEY.md` section 8f): integer witness columns,
//! the Spartan outer / product sums, the sparse read-write matrix and the instruction read-RAF scans.  Each is what the corresponding
//! kernel of `jolt_kernels::optimized` calls in place of its rayon loop; host-side O(rounds) / O(256) work stays in that kernel.
use std::ptr;
use std::sync::Arc;

use jolt_field::Fr;

use crate::context::{HipContext, HipTable};
use crate::ffi;
use crate::status::{check, HipError};

/// Device-resident machine integers: the compact scalars of `Polynomial<T>` (`crates/jolt-poly/src/dense.rs:129-142`).
pub struct HipInts {
    ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_ints,
}
// SAFETY: see HipContext.
unsafe impl Send for HipInts {}

impl HipInts {
    pub fn from_u64(ctx: &Arc<HipContext>, values: &[u64]) -> Result<Self, HipError> {
        Self::upload(ctx, values.as_ptr().cast(), ffi::JOLT_INT_U64, values.len())
    }
    pub fn from_i64(ctx: &Arc<HipContext>, values: &[i64]) -> Result<Self, HipError> {
        Self::upload(ctx, values.as_ptr().cast(), ffi::JOLT_INT_I64, values.len())
    }
    pub fn from_i128(ctx: &Arc<HipContext>, values: &[i128]) -> Result<Self, HipError> {
        Self::upload(ctx, values.as_ptr().cast(), ffi::JOLT_INT_I128, values.len())
    }
    /// Adopt a handle the library returned (`jolt_ints_from_rows`).
    pub(crate) fn from_raw(ctx: &Arc<HipContext>, raw: *mut ffi::jolt_ints) -> Self {
        Self { ctx: Arc::clone(ctx), raw }
    }
    fn upload(ctx: &Arc<HipContext>, host: *const core::ffi::c_void, kind: i32, count: usize) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: `host` points at `count` little-endian integers of the stated width; the upload is synchronous.
        check(unsafe { ffi::jolt_ints_upload(ctx.raw, host, kind, count, &mut raw) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw })
    }
    /// `Polynomial::bind_to_field`'s promotion of a window of the column.
    pub fn to_table(&self, offset: usize, len: usize) -> Result<HipTable, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: live handles, valid out-pointer.
        check(unsafe { ffi::jolt_table_from_ints(self.ctx.raw, self.raw, offset, len, &mut raw) }, self.ctx.raw)?;
        Ok(HipTable { ctx: Arc::clone(&self.ctx), raw })
    }
}
impl Drop for HipInts {
    fn drop(&mut self) {
        // SAFETY: created by jolt_ints_upload.
        let _ = unsafe { ffi::jolt_ints_free(self.ctx.raw, self.raw) };
    }
}

fn raw_ints(columns: &[&HipInts]) -> Vec<*const ffi::jolt_ints> {
    columns.iter().map(|c| c.raw.cast_const()).collect()
}

/// Stage-1 / stage-2 Spartan sums off the typed integer columns (`optimized/spartan_outer.rs:276-370,780-850`,
/// `optimized/spartan_product.rs:86-230,321-437`).  `streams` = 2 for outer (cycle || stream), 1 for product virtualization.
pub struct SpartanSums<'a> {
    pub ctx: &'a Arc<HipContext>,
    pub inputs: &'a [&'a HipInts],
    pub streams: u32,
}

impl SpartanSums<'_> {
    /// `t1(node)` for every extended node; `a` / `b`: integer column weights `[node][stream][1 + inputs]`
    /// (the integer Lagrange extension coefficients folded over the constraint rows).
    pub fn uniskip_sums(&self, eq: &HipTable, a: &[i64], b: &[i64], nodes: usize) -> Result<Vec<Fr>, HipError> {
        debug_assert_eq!(a.len(), nodes * self.streams as usize * (1 + self.inputs.len()));
        let cols = raw_ints(self.inputs);
        let mut out = vec![Fr::default(); nodes];
        // SAFETY: array lengths as asserted; layouts as in context.rs.
        check(
            unsafe {
                ffi::jolt_r1cs_uniskip_sums_small(self.ctx.raw, cols.as_ptr(), cols.len(), eq.raw, self.streams, a.as_ptr(), b.as_ptr(), nodes, out.as_mut_ptr().cast())
            },
            self.ctx.raw,
        )?;
        Ok(out)
    }

    /// The remainder's bound Az / Bz (left / right) tables under the uni-skip challenge's Lagrange weights (`fold_group`, `cell()`).
    pub fn materialize(&self, a: &[Fr], b: &[Fr]) -> Result<(HipTable, HipTable), HipError> {
        let cols = raw_ints(self.inputs);
        let (mut az, mut bz) = (ptr::null_mut(), ptr::null_mut());
        // SAFETY: as above.
        check(
            unsafe { ffi::jolt_r1cs_materialize_small(self.ctx.raw, cols.as_ptr(), cols.len(), self.streams, a.as_ptr().cast(), b.as_ptr().cast(), &mut az, &mut bz) },
            self.ctx.raw,
        )?;
        Ok((HipTable { ctx: Arc::clone(self.ctx), raw: az }, HipTable { ctx: Arc::clone(self.ctx), raw: bz }))
    }

    /// `compute_claimed_inputs`: every input evaluated at `r_cycle` from one eq table.
    pub fn claimed_inputs(&self, r_cycle: &[Fr]) -> Result<Vec<Fr>, HipError> {
        let cols = raw_ints(self.inputs);
        let mut out = vec![Fr::default(); cols.len()];
        // SAFETY: as above.
        check(unsafe { ffi::jolt_ints_evaluate(self.ctx.raw, cols.as_ptr(), cols.len(), r_cycle.as_ptr().cast(), r_cycle.len(), out.as_mut_ptr().cast()) }, self.ctx.raw)?;
        Ok(out)
    }
}

/// `CycleMajorMatrix` / `AddressMajorMatrix` of RAM read/write checking (`optimized/rw_matrix.rs`, `optimized/ram_read_write.rs:58-330`).
pub struct HipRwMatrix {
    ctx: Arc<HipContext>,
    raw: *mut ffi::jolt_rw_matrix,
}
// SAFETY: see HipContext.
unsafe impl Send for HipRwMatrix {}

impl HipRwMatrix {
    /// `addresses[j] = u64::MAX` on a cycle without RAM access (`ram_trace.rs:22`).
    #[allow(clippy::too_many_arguments)]
    pub fn new(ctx: &Arc<HipContext>, addresses: &[u64], pre: &[u64], post: &[u64], inc: &HipTable, val_init: &HipTable, tau_low: &[Fr], gamma: Fr) -> Result<Self, HipError> {
        debug_assert!(addresses.len() == pre.len() && pre.len() == post.len() && addresses.len() == 1 << tau_low.len());
        let mut raw = ptr::null_mut();
        // SAFETY: slices of equal length, tables on the same context.
        check(
            unsafe {
                ffi::jolt_rw_matrix_create(ctx.raw, addresses.as_ptr(), pre.as_ptr(), post.as_ptr(), addresses.len(), inc.raw, val_init.raw, tau_low.as_ptr().cast(),
                                           (&gamma as *const Fr).cast(), &mut raw)
            },
            ctx.raw,
        )?;
        Ok(Self { ctx: Arc::clone(ctx), raw })
    }
    /// The same member over access columns that are already resident in HBM (uploaded once per trace): no host pass, no upload per proof.
    #[allow(clippy::too_many_arguments)]
    pub fn new_resident(ctx: &Arc<HipContext>, addresses: &HipInts, pre: &HipInts, post: &HipInts, inc: &HipTable, val_init: &HipTable, tau_low: &[Fr], gamma: Fr) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: live handles of one context; the columns are u64 `jolt_ints` (checked by the library).
        check(
            unsafe {
                ffi::jolt_rw_matrix_create_resident(ctx.raw, addresses.raw, pre.raw, post.raw, inc.raw, val_init.raw, tau_low.as_ptr().cast(), (&gamma as *const Fr).cast(), &mut raw)
            },
            ctx.raw,
        )?;
        Ok(Self { ctx: Arc::clone(ctx), raw })
    }
    /// One round: the two sums of the message -- `(q(0), q(inf))` in the cycle rounds, `(s(0), s(2))` in the address rounds -- and
    /// `{current_scalar, tau_low[current_index - 1], 0}` for `gruen_poly_deg_3`.
    pub fn prove_round(&mut self, bind: Option<Fr>) -> Result<([Fr; 2], [Fr; 3]), HipError> {
        let (mut evals, mut aux) = ([Fr::default(); 2], [Fr::default(); 3]);
        let bind_ptr = bind.as_ref().map_or(ptr::null(), |b| (b as *const Fr).cast());
        // SAFETY: out-arrays of the documented size.
        check(unsafe { ffi::jolt_rw_matrix_prove_round(self.raw, bind_ptr, evals.as_mut_ptr().cast(), aux.as_mut_ptr().cast()) }, self.ctx.raw)?;
        Ok((evals, aux))
    }
    pub fn finish(&mut self, bind: Fr) -> Result<(), HipError> {
        // SAFETY: live handle.
        check(unsafe { ffi::jolt_rw_matrix_finish(self.raw, (&bind as *const Fr).cast()) }, self.ctx.raw)
    }
    /// `{ra, val, inc, bound cycle-eq factor}` at the bound point: `RamReadWriteOutputClaims` + `validate_derived_tables`.
    pub fn final_values(&mut self) -> Result<[Fr; 4], HipError> {
        let mut out = [Fr::default(); 4];
        // SAFETY: four elements as documented.
        check(unsafe { ffi::jolt_rw_matrix_final_values(self.raw, out.as_mut_ptr().cast()) }, self.ctx.raw)?;
        Ok(out)
    }
}
/// The single row a rank's LOCAL matrix is left with after its local cycle rounds (`jolt_rw_matrix_export_row`): what the ranks of a sharded prover exchange.
#[derive(Clone, Debug, Default)]
pub struct RwRow {
    pub cols: Vec<u64>,
    pub prev: Vec<u64>,
    pub next: Vec<u64>,
    pub val: Vec<Fr>,
    pub ra: Vec<Fr>,
    /// registers matrices only (empty for RAM)
    pub wa: Vec<Fr>,
    /// the rank's bound increment `inc(r_local)`
    pub inc: Fr,
    /// the rank's bound cycle-eq factor
    pub scalar: Fr,
}

impl HipRwMatrix {
    /// Hypercube-sharded provers (one process per GPU, cycles dealt in blocks; the reference is single-process): call BEFORE the local cycle rounds so that the row
    /// they leave stays in cycle-major form for [`HipRwMatrix::export_row`].
    pub fn hold_row(&mut self) -> Result<(), HipError> {
        // SAFETY: live handle.
        check(unsafe { ffi::jolt_rw_matrix_hold_row(self.raw) }, self.ctx.raw)
    }
    /// Ingest the last local challenge without asking for another round message.
    pub fn bind(&mut self, r: Fr) -> Result<(), HipError> {
        // SAFETY: live handle, one field element.
        check(unsafe { ffi::jolt_rw_matrix_bind(self.raw, (&r as *const Fr).cast()) }, self.ctx.raw)
    }
    /// The row left after the local cycle rounds: cells in column order with their raw checkpoints, the bound increment and eq factor.
    pub fn export_row(&mut self, registers: bool) -> Result<RwRow, HipError> {
        let mut cap = 0usize;
        // SAFETY: live handle, valid out-pointer.
        check(unsafe { ffi::jolt_rw_matrix_len(self.raw, &mut cap) }, self.ctx.raw)?;
        let mut row = RwRow { cols: vec![0; cap], prev: vec![0; cap], next: vec![0; cap], val: vec![Fr::default(); cap], ra: vec![Fr::default(); cap],
                              wa: if registers { vec![Fr::default(); cap] } else { Vec::new() }, ..RwRow::default() };
        let mut n = 0usize;
        let wa_ptr = if registers { row.wa.as_mut_ptr().cast() } else { ptr::null_mut() };
        // SAFETY: every array holds `cap` entries; wa may be null for a RAM matrix; the scalars are single field elements.
        check(
            unsafe {
                ffi::jolt_rw_matrix_export_row(self.raw, cap, row.cols.as_mut_ptr(), row.prev.as_mut_ptr(), row.next.as_mut_ptr(), row.val.as_mut_ptr().cast(),
                                               row.ra.as_mut_ptr().cast(), wa_ptr, (&mut row.inc as *mut Fr).cast(), (&mut row.scalar as *mut Fr).cast(), &mut n)
            },
            self.ctx.raw,
        )?;
        for v in [&mut row.cols, &mut row.prev, &mut row.next] {
            v.truncate(n);
        }
        row.val.truncate(n);
        row.ra.truncate(n);
        row.wa.truncate(n);
        Ok(row)
    }
    /// The matrix of the remaining `log2(rows.len())` cycle variables from the ranks' rows in rank order (`jolt_rw_matrix_create_merged`), built identically on every
    /// rank: `w_high` = the high coordinates of the cycle point, `scalar` = the product the ranks' eq factors combine to (`eq(w_low, r_local)`), `val_init` the
    /// initial memory (RAM) or `None` (registers start at zero).  `prove_round` / `finish` / `final_values` continue on it.
    #[allow(clippy::too_many_arguments)]
    pub fn merged(ctx: &Arc<HipContext>, registers: bool, log_k: usize, rows: &[RwRow], val_init: Option<&HipTable>, w_high: &[Fr], scalar: Fr, gamma: Fr) -> Result<Self, HipError> {
        if rows.is_empty() || !rows.len().is_power_of_two() || 1usize << w_high.len() != rows.len() {
            return Err(HipError::size_mismatch("one row per rank, a power of two of them, and one high coordinate per remaining cycle variable"));
        }
        let n: usize = rows.iter().map(|r| r.cols.len()).sum();
        let (mut rr, mut cols, mut prev, mut next) = (Vec::with_capacity(n), Vec::with_capacity(n), Vec::with_capacity(n), Vec::with_capacity(n));
        let (mut val, mut ra, mut wa, mut inc) = (Vec::with_capacity(n), Vec::with_capacity(n), Vec::with_capacity(n), Vec::with_capacity(rows.len()));
        for (g, row) in rows.iter().enumerate() {
            rr.extend(std::iter::repeat(g as u64).take(row.cols.len()));
            cols.extend_from_slice(&row.cols);
            prev.extend_from_slice(&row.prev);
            next.extend_from_slice(&row.next);
            val.extend_from_slice(&row.val);
            ra.extend_from_slice(&row.ra);
            wa.extend_from_slice(&row.wa);
            inc.push(row.inc);
        }
        let mut raw = ptr::null_mut();
        let wa_ptr: *const ffi::jolt_fr_t = if registers { wa.as_ptr().cast() } else { ptr::null() };
        // SAFETY: all cell arrays hold `n` entries, `inc` one per row, `w_high` log2(rows) coordinates; handles live on `ctx`.
        check(
            unsafe {
                ffi::jolt_rw_matrix_create_merged(ctx.raw, i32::from(registers), w_high.len(), log_k, n, rr.as_ptr(), cols.as_ptr(), prev.as_ptr(), next.as_ptr(), val.as_ptr().cast(),
                                                  ra.as_ptr().cast(), wa_ptr, inc.as_ptr().cast(), val_init.map_or(ptr::null(), |t| t.raw.cast_const()), w_high.as_ptr().cast(),
                                                  (&scalar as *const Fr).cast(), (&gamma as *const Fr).cast(), &mut raw)
            },
            ctx.raw,
        )?;
        Ok(Self { ctx: Arc::clone(ctx), raw })
    }
}

impl Drop for HipRwMatrix {
    fn drop(&mut self) {
        // SAFETY: created by jolt_rw_matrix_create.
        let _ = unsafe { ffi::jolt_rw_matrix_destroy(self.raw) };
    }
}

/// Hot-index columns resident on the device (`jolt_onehot`): one byte per (column, cycle), `0xFF` on a cold cycle.
pub struct HipHotIndices {
    ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_onehot,
    n_columns: usize,
}
// SAFETY: see HipContext.
unsafe impl Send for HipHotIndices {}
impl HipHotIndices {
    /// Adopt a handle the library returned (`jolt_onehot_from_rows`).
    pub(crate) fn from_raw(ctx: &Arc<HipContext>, raw: *mut ffi::jolt_onehot, n_columns: usize) -> Self {
        Self { ctx: Arc::clone(ctx), raw, n_columns }
    }
    /// `indices[p * cycles + j]` in `[0, k)` or `0xFF`.
    pub fn upload(ctx: &Arc<HipContext>, indices: &[u8], n_columns: usize, cycles: usize, k: u32) -> Result<Self, HipError> {
        debug_assert_eq!(indices.len(), n_columns * cycles);
        let mut raw = ptr::null_mut();
        // SAFETY: `indices` holds n_columns * cycles bytes; the upload is synchronous.
        check(unsafe { ffi::jolt_onehot_upload(ctx.raw, indices.as_ptr(), n_columns, cycles, k, &mut raw) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw, n_columns })
    }

    /// The same for address spaces beyond 255 (`jolt_onehot_upload16`: two bytes per entry, `0xFFFF` = cold cycle; `log_k_chunk = 8` from
    /// `log T >= 25`, `crates/jolt-prover/src/config.rs:175-186`).
    pub fn upload16(ctx: &Arc<HipContext>, indices: &[u16], n_columns: usize, cycles: usize, k: u32) -> Result<Self, HipError> {
        debug_assert_eq!(indices.len(), n_columns * cycles);
        let mut raw = ptr::null_mut();
        // SAFETY: `indices` holds n_columns * cycles u16 entries; the upload is synchronous.
        check(unsafe { ffi::jolt_onehot_upload16(ctx.raw, indices.as_ptr(), n_columns, cycles, k, &mut raw) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw, n_columns })
    }

    /// Per residue class `c` of the cycle mod `2^shift` and per column the sum of the bases at `(hot * T + j) >> shift` (`jolt_grid_commit_onehot_classes`): result
    /// `[c * n_columns + p]`.  The commitments of the joint polynomial's first folds are linear combinations of these (`HipHyperKzg::open_resident_with_levels`).
    pub fn grid_commit_classes(&self, srs: &crate::msm::HipSrs, shift: u32) -> Result<Vec<jolt_crypto::Bn254G1>, HipError> {
        let mut out = vec![jolt_crypto::Bn254G1::default(); self.n_columns << shift];
        // SAFETY: live handles of one context; `out` holds 2^shift * n_columns points.
        check(unsafe { ffi::jolt_grid_commit_onehot_classes(self.ctx.raw, srs.raw, self.raw, shift, out.as_mut_ptr().cast()) }, self.ctx.raw)?;
        Ok(out)
    }

    /// `kzg_commit` of every column as a 0/1 polynomial on the K x T commitment grid (`jolt_grid_commit_onehot`: the sum of the T bases a
    /// column selects, no scalars; `crates/jolt-hyperkzg/src/kzg.rs:15-27` over `TracePlacement` cycle-major, `optimized/opening.rs:340-372`).
    pub fn grid_commit(&self, srs: &crate::msm::HipSrs) -> Result<Vec<jolt_crypto::Bn254G1>, HipError> {
        let mut out = vec![jolt_crypto::Bn254G1::default(); self.n_columns];
        // SAFETY: live handles of one context; `out` holds one jolt_g1_t per column.
        check(unsafe { ffi::jolt_grid_commit_onehot(self.ctx.raw, srs.raw, self.raw, out.as_mut_ptr().cast()) }, self.ctx.raw)?;
        Ok(out)
    }
}
impl Drop for HipHotIndices {
    fn drop(&mut self) {
        // SAFETY: owned handle of a live context.
        let _ = unsafe { ffi::jolt_onehot_free(self.ctx.raw, self.raw) };
    }
}

/// The sparse cycle-major matrix of registers read/write checking (`optimized/registers_read_write/{mod,sparse,rows}.rs`): what
/// `OptimizedRegistersReadWrite::prepare` builds (`mod.rs:79-172`) and `ReadWriteKernel`'s `ProveRounds` drives (`mod.rs:374-394`).
pub struct HipRegistersRw {
    ctx: Arc<HipContext>,
    raw: *mut ffi::jolt_rw_matrix,
}
// SAFETY: see HipContext.
unsafe impl Send for HipRegistersRw {}

impl HipRegistersRw {
    /// The sharded form (see [`HipRwMatrix::hold_row`]): the same handle type underneath, so a rank's local registers matrix is held, bound and exported through a
    /// borrowed [`HipRwMatrix`] view; the merged matrix is `HipRwMatrix::merged(ctx, true, ..)`, driven by `jolt_registers_rw_prove_round` through
    /// [`HipRegistersRw::from_merged`].
    pub fn sharded<R>(&mut self, f: impl FnOnce(&mut HipRwMatrix) -> Result<R, HipError>) -> Result<R, HipError> {
        let mut view = std::mem::ManuallyDrop::new(HipRwMatrix { ctx: Arc::clone(&self.ctx), raw: self.raw });  // not dropped: `self` owns the handle
        f(&mut view)
    }
    /// Adopt a merged matrix built with `registers = true`.
    pub fn from_merged(merged: HipRwMatrix) -> Self {
        let merged = std::mem::ManuallyDrop::new(merged);
        Self { ctx: Arc::clone(&merged.ctx), raw: merged.raw }
    }
    /// `regs`: the columns rs1, rs2, rd of `RegisterCycleRow` as hot indices (k = 2^REGISTER_ADDRESS_BITS); the four value columns as u64;
    /// `inc` = RdInc.  `r_cycle` is `inputs.points.rd_write_value` (`mod.rs:110`).
    #[allow(clippy::too_many_arguments)]
    pub fn new(ctx: &Arc<HipContext>, regs: &HipHotIndices, rs1_val: &HipInts, rs2_val: &HipInts, rd_pre: &HipInts, rd_post: &HipInts, inc: &HipTable, r_cycle: &[Fr],
               gamma: Fr) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: live handles of one context.
        check(
            unsafe {
                ffi::jolt_registers_rw_create(ctx.raw, regs.raw, rs1_val.raw, rs2_val.raw, rd_pre.raw, rd_post.raw, inc.raw, r_cycle.as_ptr().cast(), (&gamma as *const Fr).cast(), &mut raw)
            },
            ctx.raw,
        )?;
        Ok(Self { ctx: Arc::clone(ctx), raw })
    }
    /// Cycle rounds: `[q(0), leading coefficient, 0, 0]` + `{current_scalar, r_cycle[current_index - 1], 0}` for `gruen_poly_deg_3`;
    /// address rounds: `[s(0), s(1), s(2), s(3)]` for `UnivariatePoly::from_evals` (`mod.rs:205-252`).
    pub fn prove_round(&mut self, bind: Option<Fr>) -> Result<([Fr; 4], [Fr; 3]), HipError> {
        let (mut evals, mut aux) = ([Fr::default(); 4], [Fr::default(); 3]);
        let bind_ptr = bind.as_ref().map_or(ptr::null(), |b| (b as *const Fr).cast());
        // SAFETY: out-arrays of the documented size.
        check(unsafe { ffi::jolt_registers_rw_prove_round(self.raw, bind_ptr, evals.as_mut_ptr().cast(), aux.as_mut_ptr().cast()) }, self.ctx.raw)?;
        Ok((evals, aux))
    }
    pub fn finish(&mut self, bind: Fr) -> Result<(), HipError> {
        // SAFETY: live handle.
        check(unsafe { ffi::jolt_rw_matrix_finish(self.raw, (&bind as *const Fr).cast()) }, self.ctx.raw)
    }
    /// `{registers_val, rd_wa, gamma * rs1_ra + gamma^2 * rs2_ra, rd_inc, bound cycle-eq factor}` (`RegistersReadWriteOutputClaims`, `mod.rs:386-402`).
    pub fn final_values(&mut self) -> Result<[Fr; 5], HipError> {
        let mut out = [Fr::default(); 5];
        // SAFETY: five elements as documented.
        check(unsafe { ffi::jolt_registers_rw_final_values(self.raw, out.as_mut_ptr().cast()) }, self.ctx.raw)?;
        Ok(out)
    }
}
impl Drop for HipRegistersRw {
    fn drop(&mut self) {
        // SAFETY: created by jolt_registers_rw_create.
        let _ = unsafe { ffi::jolt_rw_matrix_destroy(self.raw) };
    }
}

/// Rows of a resident key column (bytecode PCs, RAM word addresses) sorted by key, once per proof and column -- what the reference shares through its
/// `ProofSession` as the packed `PcRow`s (`optimized/bytecode_read_raf.rs:92-150`) and `RamAccessColumns` (`optimized/ram_trace.rs`).  Serves the pushforwards
/// of cycle weights onto the K-entry address domain: `stage_pushforwards` (`bytecode_read_raf.rs:152-237`), `fold_cycles` (`ram_trace.rs:150-162`), and the
/// final-memory column of the RAM output check.
pub struct HipKeyIndex {
    ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_key_index,
}
// SAFETY: see HipContext.
unsafe impl Send for HipKeyIndex {}

impl HipKeyIndex {
    /// `keys`: one u64 per cycle; a key `>= k` is a cold cycle (`NO_ACCESS`, an unmapped PC) and contributes nothing.
    pub fn new(ctx: &Arc<HipContext>, keys: &HipInts, k: u64) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: live handles of one context.
        check(unsafe { ffi::jolt_key_index_create(ctx.raw, keys.raw, k, &mut raw) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw })
    }

    /// `out[s][a] = sum of weights[s] over the cycles with key a`, for up to 8 weight tables in one pass (the five stage eq tables of the bytecode
    /// address phase; `eq(tau_low)` of RAM RAF evaluation).
    pub fn pushforward(&self, weights: &[&HipTable]) -> Result<Vec<HipTable>, HipError> {
        let handles: Vec<*mut ffi::jolt_table> = weights.iter().map(|t| t.raw).collect();
        let mut out: Vec<*mut ffi::jolt_table> = vec![ptr::null_mut(); weights.len()];
        // SAFETY: `handles` / `out` hold weights.len() pointers; the library fills `out` only on success.
        check(unsafe { ffi::jolt_key_index_pushforward(self.ctx.raw, self.raw, handles.as_ptr(), handles.len(), out.as_mut_ptr()) }, self.ctx.raw)?;
        Ok(out.into_iter().map(|raw| HipTable { ctx: Arc::clone(&self.ctx), raw }).collect())
    }

    /// The value at the latest cycle of every key as a field element, `init` where a key never occurs (`ram_val_final`).
    pub fn last_value(&self, values: &HipInts, init: &HipTable) -> Result<HipTable, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: live handles of one context.
        check(unsafe { ffi::jolt_key_index_last_value(self.ctx.raw, self.raw, values.raw, init.raw, &mut raw) }, self.ctx.raw)?;
        Ok(HipTable { ctx: Arc::clone(&self.ctx), raw })
    }
}
impl Drop for HipKeyIndex {
    fn drop(&mut self) {
        // SAFETY: created by jolt_key_index_create.
        let _ = unsafe { ffi::jolt_key_index_destroy(self.ctx.raw, self.raw) };
    }
}

/// The packed rows of instruction read+RAF checking on the device (`InstructionCycleRow`, `optimized/instruction_read_raf.rs:86-125`).
pub struct HipReadRaf {
    ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_read_raf,
    n_tables: u32,
}
// SAFETY: see HipContext.
unsafe impl Send for HipReadRaf {}

/// One phase's accumulators: `raf[q][chunk]` with q = left, right, identity, shift_half, shift_full, upper_all_ones, and
/// `suffix[(offset of the table + s)][chunk]`.
pub struct PhaseScan {
    pub raf: Vec<Fr>,
    pub suffix: Vec<Fr>,
}

impl HipReadRaf {
    /// `table_index[j] = 0xFF` for a cycle without a lookup table.
    pub fn new(ctx: &Arc<HipContext>, lookup_index: &[u128], table_index: &[u8], raf_flag: &[bool], n_tables: u32) -> Result<Self, HipError> {
        let flags: Vec<u8> = raf_flag.iter().map(|&f| u8::from(f)).collect();
        let mut raw = ptr::null_mut();
        // SAFETY: u128 is two little-endian u64 words on every target the prover runs on; slices of equal length.
        check(
            unsafe { ffi::jolt_read_raf_create(ctx.raw, lookup_index.as_ptr().cast(), table_index.as_ptr(), flags.as_ptr(), lookup_index.len(), n_tables, &mut raw) },
            ctx.raw,
        )?;
        Ok(Self { ctx: Arc::clone(ctx), raw, n_tables })
    }
    /// `init_phase`'s scans (`:770-812`, `:901-971`).  `suffix_kinds[t]` = `table.suffixes()` as `Suffixes as u8`.
    pub fn phase_scan(&mut self, u: &HipTable, suffix_len: u32, address_bits: u32, canonical: bool, suffix_kinds: &[Vec<u8>]) -> Result<PhaseScan, HipError> {
        debug_assert_eq!(suffix_kinds.len(), self.n_tables as usize);
        let mut offsets = Vec::with_capacity(suffix_kinds.len() + 1);
        let mut kinds = Vec::new();
        offsets.push(0u32);
        for list in suffix_kinds {
            kinds.extend_from_slice(list);
            offsets.push(kinds.len() as u32);
        }
        let mut scan = PhaseScan { raf: vec![Fr::default(); 6 * 256], suffix: vec![Fr::default(); kinds.len() * 256] };
        // SAFETY: arrays sized as the header documents.
        check(
            unsafe {
                ffi::jolt_read_raf_phase_scan(self.ctx.raw, self.raw, u.raw, suffix_len, address_bits, i32::from(canonical), offsets.as_ptr(), kinds.as_ptr(),
                                              scan.raf.as_mut_ptr().cast(), scan.suffix.as_mut_ptr().cast())
            },
            self.ctx.raw,
        )?;
        Ok(scan)
    }
    /// Condensation (`:750-758`): `u[j] *= v_prev[(lookup_index[j] >> shift) & 255]`.
    pub fn condense(&mut self, u: &mut HipTable, v_prev: &[Fr; 256], shift: u32) -> Result<(), HipError> {
        // SAFETY: 256 elements.
        check(unsafe { ffi::jolt_read_raf_condense(self.ctx.raw, self.raw, u.raw, v_prev.as_ptr().cast(), shift) }, self.ctx.raw)
    }
    /// The combined-value column and the `ra_i` columns of the cycle rounds (`pending_combined_base` / `pending_ra_base`).
    pub fn cycle_tables(&mut self, table_values: &[Fr], raf_interleaved: Fr, raf_identity: Fr, v_tables: &[Fr], address_bits: u32, ra_count: u32) -> Result<(HipTable, Vec<HipTable>), HipError> {
        let phases = (v_tables.len() / 256) as u32;
        let mut combined = ptr::null_mut();
        let mut ra = vec![ptr::null_mut(); ra_count as usize];
        // SAFETY: arrays sized as the header documents.
        check(
            unsafe {
                ffi::jolt_read_raf_cycle_tables(self.ctx.raw, self.raw, table_values.as_ptr().cast(), (&raf_interleaved as *const Fr).cast(), (&raf_identity as *const Fr).cast(),
                                                v_tables.as_ptr().cast(), phases, address_bits, ra_count, &mut combined, ra.as_mut_ptr())
            },
            self.ctx.raw,
        )?;
        let wrap = |raw| HipTable { ctx: Arc::clone(&self.ctx), raw };
        Ok((wrap(combined), ra.into_iter().map(wrap).collect()))
    }
}
impl Drop for HipReadRaf {
    fn drop(&mut self) {
        // SAFETY: created by jolt_read_raf_create.
        let _ = unsafe { ffi::jolt_read_raf_destroy(self.ctx.raw, self.raw) };
    }
}

/// The host half of `OptimizedInstructionReadRafKernel`'s 128 address rounds (`jolt_host_read_raf_address_*`): the 256-entry prefix polynomials of a phase,
/// the tables' `combine`, the RAF decompositions, checkpoints.  Together with [`HipReadRaf`] this is the kernel's `ProveRounds` body for rounds
/// `0 .. address_bits`:
///
/// ```text
/// prove_round(bind, round, previous_claim):                    // instruction_read_raf.rs:1352-1366
///     if let Some(r) = bind { if address.bind(r)? { /* phase closed */ } }
///     if round % 8 == 0 {                                       // init_phase (:747-900)
///         let p = round / 8;
///         if p > 0 { rows.condense(&mut u, &address.v_table(p - 1)?, 128 - 8 * p)?; }
///         let scan = rows.phase_scan(&u, 128 - 8 * (p + 1), 128, CANONICAL_INSTRUCTION_ADDRESS, &suffix_kinds)?;
///         address.init_phase(p, &scan)?;
///     }
///     UnivariatePoly::from_evals(&address.message(previous_claim)?)
/// ```
///
/// after round 127's bind: `address.finish()` gives the arguments of [`HipReadRaf::cycle_tables`].  No device work happens here; table ids are
/// `LookupTableKind::index()`.
pub struct HipReadRafAddress {
    raw: *mut ffi::jolt_read_raf_address,
}
// SAFETY: the handle owns plain host memory; calls are serialised by &mut self.
unsafe impl Send for HipReadRafAddress {}

impl HipReadRafAddress {
    /// `LookupTableKind::suffixes()` of all 42 tables as the device scan wants them (`Suffixes as u8`), in `LookupTableKind` order.
    pub fn suffix_kinds() -> Result<Vec<Vec<u8>>, HipError> {
        let mut offsets = [0u32; 43];
        check(unsafe { ffi::jolt_lookup_suffix_layout(offsets.as_mut_ptr(), ptr::null_mut()) }, ptr::null())?;
        let mut kinds = vec![0u8; offsets[42] as usize];
        // SAFETY: sized from the first call.
        check(unsafe { ffi::jolt_lookup_suffix_layout(offsets.as_mut_ptr(), kinds.as_mut_ptr()) }, ptr::null())?;
        Ok((0..42).map(|t| kinds[offsets[t] as usize..offsets[t + 1] as usize].to_vec()).collect())
    }
    /// `table_present[t]`: some cycle's lookup table is `t` (the kernel's non-empty buckets, `:683-697`).
    pub fn new(gamma: Fr, table_present: &[bool; 42], canonical: bool) -> Result<Self, HipError> {
        let present: Vec<u8> = table_present.iter().map(|&p| u8::from(p)).collect();
        let mut raw = ptr::null_mut();
        // SAFETY: 42 flags.
        check(unsafe { ffi::jolt_host_read_raf_address_create((&gamma as *const Fr).cast(), present.as_ptr(), i32::from(canonical), &mut raw) }, ptr::null())?;
        Ok(Self { raw })
    }
    pub fn init_phase(&mut self, phase: u32, scan: &PhaseScan) -> Result<(), HipError> {
        // SAFETY: the scan of the 42-table layout.
        check(unsafe { ffi::jolt_host_read_raf_address_init_phase(self.raw, phase, scan.raf.as_ptr().cast(), scan.suffix.as_ptr().cast()) }, ptr::null())
    }
    /// `address_message` (`:973-1050`): `[s(0), s(1), s(2)]` for `UnivariatePoly::from_evals`.
    pub fn message(&mut self, previous_claim: Fr) -> Result<[Fr; 3], HipError> {
        let mut evals = [Fr::default(); 3];
        // SAFETY: three elements out.
        check(unsafe { ffi::jolt_host_read_raf_address_message(self.raw, (&previous_claim as *const Fr).cast(), evals.as_mut_ptr().cast()) }, ptr::null())?;
        Ok(evals)
    }
    /// The address branch of `bind` (`:1235-1282`); `true` when the phase closed (its eq table and the new checkpoints are in place).
    pub fn bind(&mut self, challenge: Fr) -> Result<bool, HipError> {
        let mut done = 0i32;
        check(unsafe { ffi::jolt_host_read_raf_address_bind(self.raw, (&challenge as *const Fr).cast(), &mut done) }, ptr::null())?;
        Ok(done != 0)
    }
    /// `eq(phase challenges, .)` of a closed phase: the `v_prev` of the next condensation and a row of `cycle_tables`' `v_tables`.
    pub fn v_table(&self, phase: u32) -> Result<[Fr; 256], HipError> {
        let mut out = [Fr::default(); 256];
        check(unsafe { ffi::jolt_host_read_raf_address_v_table(self.raw, phase, out.as_mut_ptr().cast()) }, ptr::null())?;
        Ok(out)
    }
    /// `init_cycle_rounds` (`:1140-1160`): `(table_values[42], raf_interleaved, raf_identity)`.
    pub fn finish(&self) -> Result<(Vec<Fr>, Fr, Fr), HipError> {
        let mut values = vec![Fr::default(); 42];
        let (mut interleaved, mut identity) = (Fr::default(), Fr::default());
        check(
            unsafe { ffi::jolt_host_read_raf_address_finish(self.raw, values.as_mut_ptr().cast(), (&mut interleaved as *mut Fr).cast(), (&mut identity as *mut Fr).cast()) },
            ptr::null(),
        )?;
        Ok((values, interleaved, identity))
    }
}
impl Drop for HipReadRafAddress {
    fn drop(&mut self) {
        // SAFETY: created by jolt_host_read_raf_address_create.
        let _ = unsafe { ffi::jolt_host_read_raf_address_destroy(self.raw) };
    }
}
