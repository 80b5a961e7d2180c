//! `JoltGroup::msm` for `Bn254G1` on the device (`crates/jolt-crypto/src/ec/group.rs:63-70`, `ec/bn254/mod.rs:195-212`).
//!
//! HyperKZG only ever multiplies prefixes of one long-lived `g1_powers` vector (`crates/jolt-hyperkzg/src/kzg.rs:19-26,114`), whose
//! field is `pub(crate)` (`types.rs:105-108`): an out-of-crate backend sees it only as the `bases` argument.  The device copy is
//! therefore cached by (context, `bases.as_ptr()`) with a content fingerprint (first / last base) checked on every hit: the first MSM
//! over a base vector uploads it (converted to affine ONCE, where the reference converts every base on every call, `bn254/mod.rs:205`),
//! later prefix MSMs reuse it.  Entries are released by `msm_cache_evict` / `msm_cache_clear` (call the former when the prover setup
//! that owns `g1_powers` is dropped).  A prover that can keep a handle next to its setup should hold a `HipSrs` directly instead.
use std::collections::HashMap;
use std::ptr;
use std::sync::{Arc, Mutex, OnceLock};

use jolt_crypto::Bn254G1;
use jolt_field::Fr;

use crate::context::HipContext;
use crate::ffi;
use crate::status::{check, HipError};

/// `jolt_srs`: affine bases resident in HBM (+ optional fixed-base window tables).
pub struct HipSrs {
    ctx: Arc<HipContext>,
    pub(crate) raw: *mut ffi::jolt_srs,
    len: usize,
}

// SAFETY: see HipContext.
unsafe impl Send for HipSrs {}

impl HipSrs {
    pub fn upload(ctx: &Arc<HipContext>, bases: &[Bn254G1]) -> Result<Self, HipError> {
        let mut raw = ptr::null_mut();
        // SAFETY: `Bn254G1` is `#[repr(transparent)]` over `ark_bn254::G1Projective` = three Montgomery Fq = `jolt_g1_t`
        // (crates/jolt-crypto/src/ec/bn254/mod.rs:17-24, layout assertions :21-24).
        check(unsafe { ffi::jolt_srs_upload_g1(ctx.raw, bases.as_ptr().cast(), bases.len(), &mut raw) }, ctx.raw)?;
        Ok(Self { ctx: Arc::clone(ctx), raw, len: bases.len() })
    }

    /// Window-precomputed tables for a long-lived SRS (`jolt_srs_precompute_windows`): ceil(255 / c) copies of the bases in HBM buy
    /// one shared bucket set per MSM and 24-bit windows.  Opt-in: worth it when the same SRS serves many large MSMs.
    pub fn precompute_windows(&mut self) -> Result<(), HipError> {
        // SAFETY: live handles.
        check(unsafe { ffi::jolt_srs_precompute_windows(self.ctx.raw, self.raw, 0, 0) }, self.ctx.raw)
    }

    /// `sum_i scalars[i] * bases[i]` over the prefix `bases[..scalars.len()]`.
    pub fn msm(&self, scalars: &[Fr]) -> Result<Bn254G1, HipError> {
        let mut out = Bn254G1::default();
        // SAFETY: layouts as above; `out` is one jolt_g1_t.
        check(unsafe { ffi::jolt_msm_g1(self.ctx.raw, self.raw, scalars.as_ptr().cast(), scalars.len(), (&mut out as *mut Bn254G1).cast()) }, self.ctx.raw)?;
        Ok(out)
    }

    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
}

impl Drop for HipSrs {
    fn drop(&mut self) {
        // SAFETY: owned handle of a live context.
        let _ = unsafe { ffi::jolt_srs_free(self.ctx.raw, self.raw) };
    }
}

/// The context as the MSM drop-in shares it between rayon workers.  `HipContext` is `Send` but deliberately not `Sync` (its other
/// methods assume one thread at a time); the ONLY operation reachable through a `SharedMsmContext` is `msm_g1`, which holds the
/// context's device lock from the cache lookup to the end of the device call.
pub struct SharedMsmContext(Arc<HipContext>);

// SAFETY: every use of the inner context through this wrapper happens under `HipContext::exclusive` (see `msm_g1`).
unsafe impl Send for SharedMsmContext {}
unsafe impl Sync for SharedMsmContext {}

impl SharedMsmContext {
    pub fn new(ctx: &Arc<HipContext>) -> Self {
        Self(Arc::clone(ctx))
    }
}

/// What identifies a cached base vector: the context it was uploaded to, where the host slice starts, and a fingerprint of its
/// content (first and last base, 96 bytes each) so that a NEW vector that happens to land at a freed vector's address (a second setup
/// with another beta, tests in one process) is not served from the stale device copy.
#[derive(Clone, PartialEq, Eq, Hash)]
struct SrsKey {
    ctx: usize,
    ptr: usize,
}

struct SrsEntry {
    len: usize,
    first: [u8; 96],
    last: [u8; 96],
    /// `HipSrs` is `Send` only (one context, one thread at a time): the `Mutex` is what makes the cached handle shareable, and it is
    /// held across the whole device MSM -- see `msm_g1`.
    srs: Arc<Mutex<HipSrs>>,
}

fn base_bytes(b: &Bn254G1) -> [u8; 96] {
    let mut out = [0u8; 96];
    // SAFETY: `Bn254G1` is `#[repr(transparent)]` over three Montgomery Fq = 96 plain bytes (crates/jolt-crypto/src/ec/bn254/mod.rs:17-24).
    unsafe { ptr::copy_nonoverlapping((b as *const Bn254G1).cast::<u8>(), out.as_mut_ptr(), 96) };
    out
}

fn cache() -> &'static Mutex<HashMap<SrsKey, SrsEntry>> {
    static CACHE: OnceLock<Mutex<HashMap<SrsKey, SrsEntry>>> = OnceLock::new();
    CACHE.get_or_init(|| Mutex::new(HashMap::new()))
}

/// Releases the device copy (bases + window tables) of a base vector, e.g. from the `Drop` of the prover setup that owns `g1_powers`.
/// Without it an entry lives until `msm_cache_clear` or until a different vector is seen at the same address.
pub fn msm_cache_evict(ctx: &SharedMsmContext, bases: &[Bn254G1]) {
    let ctx = &ctx.0;
    if let Ok(mut map) = cache().lock() {
        let _ = map.remove(&SrsKey { ctx: ctx.raw as usize, ptr: bases.as_ptr() as usize });
    }
}

/// Drops every cached device SRS (all contexts).
pub fn msm_cache_clear() {
    if let Ok(mut map) = cache().lock() {
        map.clear();
    }
}

/// Drop-in body for `impl JoltGroup for Bn254G1 { fn msm(..) }`: same panic on a length mismatch, the device result on success,
/// `None` when the device path is unavailable (the caller keeps arkworks' `VariableBaseMSM`).
///
/// Prefix reuse: a base slice that starts where a cached vector of the SAME context starts, is no longer than it and has the same
/// first base is served from that vector (`kzg_commit` / `kzg_open_batch` only ever pass prefixes of `g1_powers`).
///
/// Concurrency: the reference calls `msm` from rayon workers (`HyperKZGScheme::open` commits the levels with
/// `polys.par_iter().skip(1).map(kzg_commit)`, scheme.rs:141-145), while a `jolt_ctx` is used by one thread at a time (its MSM lanes,
/// workspace and pinned result buffer are per context).  Device MSMs are therefore serialised per context: the context's lock is held
/// from the cache lookup to the end of `jolt_msm_g1`.  The MSM itself fills the GPU; concurrent callers only queue.
#[must_use]
pub fn msm_g1(ctx: &SharedMsmContext, bases: &[Bn254G1], scalars: &[Fr]) -> Option<Bn254G1> {
    let ctx = &ctx.0;
    assert_eq!(bases.len(), scalars.len(), "msm: bases/scalars length mismatch"); // group.rs:66-69
    if bases.is_empty() {
        return Some(Bn254G1::default());
    }
    let _device = ctx.exclusive(); // one device call at a time per context
    let key = SrsKey { ctx: ctx.raw as usize, ptr: bases.as_ptr() as usize };
    let first = base_bytes(&bases[0]);
    let srs = {
        let mut map = cache().lock().ok()?;
        let usable = match map.get(&key) {
            // a prefix of the cached vector: same start, not longer, same first base -- and, when the lengths agree, the same last base
            Some(e) => e.first == first && (bases.len() < e.len || (bases.len() == e.len && e.last == base_bytes(&bases[bases.len() - 1]))),
            None => false,
        };
        if !usable {
            // unseen, longer than the cached copy, or a different vector at a recycled address: (re)upload; the old entry is released
            let fresh = HipSrs::upload(ctx, bases).ok()?;
            let _ = map.insert(
                key.clone(),
                SrsEntry { len: bases.len(), first, last: base_bytes(&bases[bases.len() - 1]), srs: Arc::new(Mutex::new(fresh)) },
            );
        }
        Arc::clone(&map.get(&key)?.srs)
    };
    let guard = srs.lock().ok()?;
    guard.msm(scalars).ok()
}

/// One rank's part of a HyperKZG prover sharded over the `world` GPUs of a node (`DESIGN.md` section 6; the reference is single
/// process).  The rank holds the bases of the indices it owns under the subtree assignment, compacted in index order, with their
/// window tables; `HyperKZGScheme::open` (`crates/jolt-hyperkzg/src/scheme.rs:122-158`) then runs on `1 / world` of the polynomial
/// per rank and every rank returns the same proof.
pub struct HipShardedOpening {
    ctx: Arc<HipContext>,
    srs: HipSrs,
    rank: i32,
    world: i32,
}

impl HipShardedOpening {
    /// The indices this rank owns of `[0, len)`, in the order of its compact arrays (slot -> index): which bases to upload and which
    /// evaluations to send to this GPU.
    pub fn owned_indices(len: usize, rank: i32, world: i32) -> Result<Vec<usize>, HipError> {
        let mut count = 0usize;
        // SAFETY: host-only entry points writing one usize.
        check(unsafe { ffi::jolt_host_subtree_owned_terms(len, rank, world, &mut count) }, ptr::null())?;
        (0..count)
            .map(|slot| {
                let mut index = 0usize;
                check(unsafe { ffi::jolt_host_subtree_term_index(slot, rank, world, &mut index) }, ptr::null()).map(|()| index)
            })
            .collect()
    }

    /// Uploads this rank's bases out of the full `g1_powers` and builds the window tables over them.
    pub fn new(ctx: &Arc<HipContext>, g1_powers: &[Bn254G1], rank: i32, world: i32) -> Result<Self, HipError> {
        let own: Vec<Bn254G1> = Self::owned_indices(g1_powers.len(), rank, world)?.into_iter().map(|i| g1_powers[i]).collect();
        let mut srs = HipSrs::upload(ctx, &own)?;
        srs.precompute_windows()?;
        Ok(Self { ctx: Arc::clone(ctx), srs, rank, world })
    }

    /// `open` over this rank's compact array of the evaluations (`evals[slot] = poly[owned_indices[slot]]`).  `gather` moves
    /// `count` 32-byte words from every rank to every rank (rank order): the same hook as the round sums of the sharded sumcheck.
    /// Returns (level commitments, witness commitments, evaluations `v[t][level]`) -- `HyperKZGProof`'s fields.
    pub fn open(
        &self,
        evals: &crate::context::HipTable,
        point: &[Fr],
        transcript_label: u64,
        gather: ffi::jolt_gather_fn,
        gather_user: *mut core::ffi::c_void,
    ) -> Result<(Vec<Bn254G1>, [Bn254G1; 3], Vec<Fr>), HipError> {
        let ell = point.len();
        let mut com = vec![Bn254G1::default(); ell.saturating_sub(1).max(1)];
        let mut w = [Bn254G1::default(); 3];
        let mut v = vec![Fr::default(); 3 * ell];
        // SAFETY: live handles; output arrays sized as the header states (ell - 1 points, 3 points, 3 * ell field elements).
        check(
            unsafe {
                ffi::jolt_host_hyperkzg_open_subtree(
                    self.ctx.raw,
                    self.srs.raw,
                    evals.raw,
                    point.as_ptr().cast(),
                    ell,
                    transcript_label,
                    self.rank,
                    self.world,
                    gather,
                    gather_user,
                    com.as_mut_ptr().cast(),
                    w.as_mut_ptr().cast(),
                    v.as_mut_ptr().cast(),
                    ptr::null_mut(),
                )
            },
            self.ctx.raw,
        )?;
        com.truncate(ell.saturating_sub(1));
        Ok((com, w, v))
    }
}
