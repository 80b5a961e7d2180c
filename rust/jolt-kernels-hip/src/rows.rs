//! The witness path of a proof: the tracer's per-cycle records cross PCIe ONCE, as packed rows in page-locked memory, and every column
//! the kernels read is extracted on the device.
//!
//! Reference side: `RowSource::rows()` and `WitnessBundle::from_row` (`crates/jolt-witness/src/consumer.rs:129-143`) hand the prover one typed record per cycle;
//! `CommittedColumnsWitness` / `InstructionCycleRow` (`crates/jolt-kernels/src/commitment.rs:25-32`) name the fields.  Here the records are written into a
//! [`HipPinnedRows`] buffer (little-endian fields at fixed offsets inside a row of `row_bytes` bytes), uploaded with [`HipRows::upload`], and turned into
//! [`HipInts`] (compact scalars, `Polynomial::bind_to_field` semantics at the first bind), [`HipHotIndices`] (the chunks of an address field as one byte per
//! polynomial and cycle, `RaChunkSelector::chunk_u128`, `crates/jolt-witness/src/witnesses/one_hot.rs:14-52`) or promoted [`HipTable`]s.
use std::ptr;
use std::sync::Arc;

use crate::context::{HipContext, HipTable};
use crate::ffi;
use crate::ops::{HipHotIndices, HipInts};
use crate::status::{check, HipError};

/// Page-locked host memory (`jolt_host_pinned_alloc`): the block the tracer fills; `HipRows::upload` from it runs at the link rate.
pub struct HipPinnedRows {
    ctx: Arc<HipContext>,
    ptr: *mut u8,
    len: usize,
    row_bytes: usize,
}
// SAFETY: plain host memory owned by this value; the context handle is only used to free it.
unsafe impl Send for HipPinnedRows {}

impl HipPinnedRows {
    pub fn new(ctx: &Arc<HipContext>, n_rows: usize, row_bytes: usize) -> Result<Self, HipError> {
        let len = n_rows.checked_mul(row_bytes).ok_or_else(|| HipError::size_mismatch("row buffer length overflows usize"))?;
        let mut p: *mut core::ffi::c_void = ptr::null_mut();
        // SAFETY: live context, valid out-pointer.
        check(unsafe { ffi::jolt_host_pinned_alloc(ctx.raw, len, &mut p) }, ctx.raw)?;
        // the block is not zeroed by the runtime: a row buffer must not leak earlier contents into fields the tracer leaves untouched
        // SAFETY: `p` points at `len` writable bytes.
        unsafe { ptr::write_bytes(p.cast::<u8>(), 0, len) };
        Ok(Self { ctx: Arc::clone(ctx), ptr: p.cast(), len, row_bytes })
    }
    pub fn row_bytes(&self) -> usize {
        self.row_bytes
    }
    pub fn n_rows(&self) -> usize {
        self.len / self.row_bytes
    }
    pub fn as_mut_slice(&mut self) -> &mut [u8] {
        // SAFETY: `ptr` points at `len` bytes owned by `self` for its whole life.
        unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
    pub fn as_slice(&self) -> &[u8] {
        // SAFETY: as above.
        unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
    }
    /// Row `j` as a mutable byte slice (the tracer writes its record's fields here).
    pub fn row_mut(&mut self, j: usize) -> &mut [u8] {
        let rb = self.row_bytes;
        &mut self.as_mut_slice()[j * rb..(j + 1) * rb]
    }
}
impl Drop for HipPinnedRows {
    fn drop(&mut self) {
        // SAFETY: `ptr` came from jolt_host_pinned_alloc of this context and is freed once.
        let _ = unsafe { ffi::jolt_host_pinned_free(self.ctx.raw, self.ptr.cast()) };
    }
}

/// The rows of one proof resident on the device.
pub struct HipRows {
    ctx: Arc<HipContext>,
    raw: *mut ffi::jolt_rows,
    n_rows: usize,
    row_bytes: usize,
}
// SAFETY: see HipContext.
unsafe impl Send for HipRows {}

impl HipRows {
    /// One host-to-device copy of the whole buffer (synchronous: the buffer may be refilled when this returns).
    pub fn upload(ctx: &Arc<HipContext>, rows: &HipPinnedRows) -> Result<Self, HipError> {
        Self::upload_bytes(ctx, rows.as_slice(), rows.row_bytes())
    }
    /// The same from ordinary (pageable) memory: the runtime stages the copy.
    pub fn upload_bytes(ctx: &Arc<HipContext>, bytes: &[u8], row_bytes: usize) -> Result<Self, HipError> {
        if row_bytes == 0 || bytes.is_empty() || bytes.len() % row_bytes != 0 {
            return Err(HipError::size_mismatch("row buffer is not a whole number of rows"));
        }
        let n_rows = bytes.len() / row_bytes;
        let mut raw = ptr::null_mut();
        // SAFETY: `bytes` holds n_rows * row_bytes bytes; the upload is synchronous.
        check(unsafe { ffi::jolt_rows_upload(ctx.raw, bytes.as_ptr().cast(), n_rows, row_bytes, &mut raw) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw, n_rows, row_bytes })
    }
    /// The copy IN FLIGHT beside whatever the context does next (`jolt_rows_upload_begin`): the NEXT proof's rows cross the link under the current proof's kernels.
    /// The buffer is borrowed until the returned [`HipRowsInFlight`] is waited for (or dropped): the tracer cannot refill it before that, which is what the
    /// lifetime says.
    pub fn begin<'a>(ctx: &Arc<HipContext>, rows: &'a HipPinnedRows) -> Result<HipRowsInFlight<'a>, HipError> {
        let bytes = rows.as_slice();
        let row_bytes = rows.row_bytes();
        if row_bytes == 0 || bytes.is_empty() || bytes.len() % row_bytes != 0 {
            return Err(HipError::size_mismatch("row buffer is not a whole number of rows"));
        }
        let n_rows = bytes.len() / row_bytes;
        let mut raw = ptr::null_mut();
        // SAFETY: `bytes` is page-locked (HipPinnedRows), holds n_rows * row_bytes bytes and stays borrowed -- hence unchanged and alive -- until the wait.
        check(unsafe { ffi::jolt_rows_upload_begin(ctx.raw, bytes.as_ptr().cast(), n_rows, row_bytes, &mut raw) }, ctx.raw)?;
        Ok(HipRowsInFlight { rows: Some(Self { ctx: Arc::clone(ctx), raw, n_rows, row_bytes }), _source: std::marker::PhantomData })
    }
    pub fn n_rows(&self) -> usize {
        self.n_rows
    }
    /// Every integer field of `fields` (`(offset, width, signed)`) as a compact integer column, all of them in ONE pass over the rows
    /// (`jolt_ints_from_rows_many`: whole rows staged in LDS) -- what a proof's witness hand-over calls instead of [`HipRows::ints`] per field.
    pub fn ints_many(&self, fields: &[(usize, u32, bool)]) -> Result<Vec<HipInts>, HipError> {
        for &(offset, width, _) in fields {
            self.field_ok(offset, width)?;
        }
        let offsets: Vec<usize> = fields.iter().map(|f| f.0).collect();
        let widths: Vec<u32> = fields.iter().map(|f| f.1).collect();
        let signed: Vec<i32> = fields.iter().map(|f| i32::from(f.2)).collect();
        let mut raws: Vec<*mut ffi::jolt_ints> = vec![ptr::null_mut(); fields.len()];
        // SAFETY: live handles of one context; the four arrays hold `fields.len()` entries each; on failure the library has released what it had created.
        check(
            unsafe { ffi::jolt_ints_from_rows_many(self.ctx.raw, self.raw, offsets.as_ptr(), widths.as_ptr(), signed.as_ptr(), fields.len(), raws.as_mut_ptr()) },
            self.ctx.raw,
        )?;
        Ok(raws.into_iter().map(|raw| HipInts::from_raw(&self.ctx, raw)).collect())
    }
    fn field_ok(&self, offset: usize, width: u32) -> Result<(), HipError> {
        if matches!(width, 1 | 2 | 4 | 8) && (width as usize) <= self.row_bytes && offset <= self.row_bytes - width as usize {
            Ok(())
        } else {
            Err(HipError::size_mismatch("field outside the row"))
        }
    }
    /// The field at `offset` (1, 2, 4 or 8 bytes, little endian) of every row as a compact integer column (`u64`, or `i64` when `signed`).
    pub fn ints(&self, offset: usize, width: u32, signed: bool) -> Result<HipInts, HipError> {
        self.field_ok(offset, width)?;
        let mut raw = ptr::null_mut();
        // SAFETY: live handles of one context, valid out-pointer.
        check(unsafe { ffi::jolt_ints_from_rows(self.ctx.raw, self.raw, offset, width, i32::from(signed), &mut raw) }, self.ctx.raw)?;
        Ok(HipInts::from_raw(&self.ctx, raw))
    }
    /// The same field promoted to field elements (`Ring::from_u64` / `from_i64` per entry).
    pub fn table(&self, offset: usize, width: u32, signed: bool) -> Result<HipTable, HipError> {
        self.field_ok(offset, width)?;
        let mut raw = ptr::null_mut();
        // SAFETY: as above.
        check(unsafe { ffi::jolt_table_from_rows(self.ctx.raw, self.raw, offset, width, i32::from(signed), &mut raw) }, self.ctx.raw)?;
        Ok(HipTable { ctx: Arc::clone(&self.ctx), raw })
    }
    /// `shifts.len()` one-hot columns from ONE address field of `width` bytes (<= 16): column i holds `(field >> shifts[i]) & (2^log_k - 1)`, or the cold
    /// sentinel where the byte at `valid_offset` is zero (`None`: every cycle is hot).
    pub fn hot_indices(&self, offset: usize, width: u32, shifts: &[u32], log_k: u32, valid_offset: Option<usize>) -> Result<HipHotIndices, HipError> {
        if width == 0 || width > 16 || (width as usize) > self.row_bytes || offset > self.row_bytes - width as usize || valid_offset.is_some_and(|v| v >= self.row_bytes) {
            return Err(HipError::size_mismatch("address field outside the row"));
        }
        let mut raw = ptr::null_mut();
        // SAFETY: live handles; `shifts` holds shifts.len() values; usize::MAX is the ABI's "no validity byte".
        check(
            unsafe {
                ffi::jolt_onehot_from_rows(self.ctx.raw, self.raw, offset, width, shifts.as_ptr(), shifts.len(), log_k, valid_offset.unwrap_or(usize::MAX), &mut raw)
            },
            self.ctx.raw,
        )?;
        Ok(HipHotIndices::from_raw(&self.ctx, raw, shifts.len()))
    }
}
impl Drop for HipRows {
    fn drop(&mut self) {
        // SAFETY: `raw` came from jolt_rows_upload of this context and is freed once.
        let _ = unsafe { ffi::jolt_rows_free(self.ctx.raw, self.raw) };
    }
}

/// Rows whose host-to-device copy was begun and not yet waited for; borrows the page-locked source for as long as the copy may read it.
pub struct HipRowsInFlight<'a> {
    rows: Option<HipRows>,
    _source: std::marker::PhantomData<&'a HipPinnedRows>,
}

impl HipRowsInFlight<'_> {
    /// Blocks the host until the copy has landed (begun a proof earlier, it has) and hands the rows over (`jolt_rows_upload_wait`).
    pub fn wait(mut self) -> Result<HipRows, HipError> {
        let rows = self.rows.take().ok_or_else(|| HipError::size_mismatch("rows already taken"))?;
        // SAFETY: live handles of one context.
        check(unsafe { ffi::jolt_rows_upload_wait(rows.ctx.raw, rows.raw) }, rows.ctx.raw)?;
        Ok(rows)
    }
}
// Dropping a `HipRowsInFlight` that was never waited for drops its `HipRows`: `jolt_rows_free` lets the copy finish before the block goes back to the pool.
