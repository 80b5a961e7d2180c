//! Raw FFI declarations of `libjolt_hip.so` -- GENERATED from `include/jolt_hip.h` by `tools/gen_rust_ffi.py`; do not edit.
//! One declaration per entry point of the C header, same order, same arity, same types (checked by tests/test_abi_cpu.py).
//! `jolt_fr_t` is bit-identical to `jolt_field::Fr` (4 x u64 Montgomery limbs, crates/jolt-field/src/bn254/mod.rs:33-43) and
//! `jolt_g1_t` to `jolt_crypto::Bn254G1` (ark_bn254::G1Projective, crates/jolt-crypto/src/ec/bn254/mod.rs:17-24).
#![allow(non_camel_case_types, clippy::too_many_arguments, clippy::missing_safety_doc)]
use core::ffi::{c_char, c_void};

pub const JOLT_HIP_ABI_VERSION: i32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct jolt_fr_t {
    pub l: [u64; 4],
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct jolt_fq_t {
    pub l: [u64; 4],
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct jolt_g1_t {
    pub x: jolt_fq_t,
    pub y: jolt_fq_t,
    pub z: jolt_fq_t,
}
#[repr(C)]
pub struct jolt_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_table {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_member {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_srs {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_ints {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_msm_pending {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_host_transcript {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_batch {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_split_lt {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_onehot {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_rows {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_read_raf {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_read_raf_address {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_rw_matrix {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_key_index {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_comm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_shm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_grid_hint {
    _private: [u8; 0],
}
#[repr(C)]
pub struct jolt_stage_op {
    _private: [u8; 0],
}

/// Status codes (`enum` of the header); see `crate::status` for the mapping onto the reference's error types.
pub const JOLT_OK: i32 = 0;
pub const JOLT_ERR_INVALID_ARG: i32 = 1;
pub const JOLT_ERR_NO_DEVICE: i32 = 2;
pub const JOLT_ERR_OOM: i32 = 3;
pub const JOLT_ERR_HIP: i32 = 4;
pub const JOLT_ERR_SIZE_MISMATCH: i32 = 5;
pub const JOLT_ERR_UNSUPPORTED: i32 = 6;
pub const JOLT_ERR_NOT_FULLY_BOUND: i32 = 7;
pub const JOLT_ERR_ROUND_CHECK: i32 = 8;
pub const JOLT_ERR_SRS_TOO_SMALL: i32 = 9;
pub const JOLT_ERR_EMPTY_POINT: i32 = 10;
pub const JOLT_ERR_NOT_INVERTIBLE: i32 = 11;
pub const JOLT_ORDER_LOW_TO_HIGH: i32 = 0;
pub const JOLT_ORDER_HIGH_TO_LOW: i32 = 1;
pub const JOLT_MEMBER_FLAG_SKIP_ONE: u32 = 1;
pub const JOLT_MEMBER_FLAG_BORROW_TABLES: u32 = 2;
pub const JOLT_INT_U64: i32 = 0;
pub const JOLT_INT_I64: i32 = 1;
pub const JOLT_INT_I128: i32 = 2;
pub const JOLT_SCALAR_FR: i32 = 3;
pub const JOLT_MAX_MEMBER_TABLES: usize = 40;
pub const JOLT_MAX_MEMBER_TERMS: usize = 16;
pub const JOLT_MAX_MEMBER_FACTORS: usize = 64;
pub const JOLT_MAX_DEGREE: usize = 7;

#[repr(C)]
pub struct jolt_member_desc {
    pub n_tables: u32,
    pub n_terms: u32,
    pub degree: u32,
    pub order: i32,
    pub term_offsets: *const u32,
    pub factors: *const u32,
    pub coeffs: *const jolt_fr_t,
}
#[repr(C)]
pub struct jolt_member_lc_desc {
    pub n_tables: u32,
    pub n_groups: u32,
    pub n_factors: u32,
    pub n_lc: u32,
    pub degree: u32,
    pub order: i32,
    pub flags: u32,
    pub group_factor_offsets: *const u32,
    pub factor_lc_offsets: *const u32,
    pub factor_consts: *const jolt_fr_t,
    pub lc_tables: *const u32,
    pub lc_coeffs: *const jolt_fr_t,
}
pub type jolt_local_round_fn = Option<
    unsafe extern "C" fn(user: *mut c_void, active: *const usize, n_active: usize, binds: *const *const jolt_fr_t, evals_out: *mut jolt_fr_t, evals_count: usize) -> i32,
>;
pub type jolt_gather_fn = Option<unsafe extern "C" fn(user: *mut c_void, local: *const jolt_fr_t, count: usize, gathered: *mut jolt_fr_t) -> i32>;
pub type jolt_round_transcript_fn = Option<unsafe extern "C" fn(user: *mut c_void, compressed_coeffs: *const jolt_fr_t, n_coeffs: usize, challenge_out: *mut jolt_fr_t) -> i32>;
pub type jolt_open_transcript_fn = Option<
    unsafe extern "C" fn(user: *mut c_void, phase: i32, points: *const jolt_g1_t, n_points: usize, values: *const jolt_fr_t, n_values: usize, challenge_out: *mut jolt_fr_t) -> i32,
>;

#[link(name = "jolt_hip")]
extern "C" {
    pub fn jolt_status_string(status: i32) -> *const c_char;
    pub fn jolt_abi_version() -> i32;
    pub fn jolt_ctx_create(device_id: i32, stream: *mut c_void, out: *mut *mut jolt_ctx) -> i32;
    pub fn jolt_ctx_destroy(ctx: *mut jolt_ctx) -> i32;
    pub fn jolt_ctx_synchronize(ctx: *mut jolt_ctx) -> i32;
    pub fn jolt_ctx_synchronize_foreground(ctx: *mut jolt_ctx) -> i32;
    pub fn jolt_ctx_bind_thread(ctx: *mut jolt_ctx) -> i32;
    pub fn jolt_last_error(ctx: *const jolt_ctx) -> *const c_char;
    pub fn jolt_ctx_trim(ctx: *mut jolt_ctx) -> i32;
    pub fn jolt_ctx_memory_stats(ctx: *const jolt_ctx, live_bytes: *mut usize, cached_bytes: *mut usize, peak_bytes: *mut usize) -> i32;
    pub fn jolt_ctx_workspace_stats(ctx: *const jolt_ctx, msm_lane_bytes: *mut usize, msm_batch_bytes: *mut usize) -> i32;
    pub fn jolt_timer_begin(ctx: *mut jolt_ctx) -> i32;
    pub fn jolt_timer_end(ctx: *mut jolt_ctx, elapsed_ms: *mut f32) -> i32;
    pub fn jolt_table_upload(ctx: *mut jolt_ctx, host: *const jolt_fr_t, len: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_table_from_device(ctx: *mut jolt_ctx, device_ptr: *const c_void, len: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_table_alloc(ctx: *mut jolt_ctx, len: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_table_clone(ctx: *mut jolt_ctx, src: *const jolt_table, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_table_from_u64(ctx: *mut jolt_ctx, host: *const u64, len: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_table_from_i64(ctx: *mut jolt_ctx, host: *const i64, len: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_table_download(ctx: *mut jolt_ctx, t: *const jolt_table, offset: usize, len: usize, host: *mut jolt_fr_t) -> i32;
    pub fn jolt_table_len(t: *const jolt_table, len: *mut usize) -> i32;
    pub fn jolt_table_device_ptr(t: *const jolt_table, device_ptr: *mut *mut c_void) -> i32;
    pub fn jolt_table_free(ctx: *mut jolt_ctx, t: *mut jolt_table) -> i32;
    pub fn jolt_table_slice(ctx: *mut jolt_ctx, parent: *const jolt_table, offset: usize, len: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_table_write(ctx: *mut jolt_ctx, t: *mut jolt_table, offset: usize, host: *const jolt_fr_t, len: usize) -> i32;
    pub fn jolt_bind(ctx: *mut jolt_ctx, tables: *const *mut jolt_table, k: usize, r: *const jolt_fr_t, order: i32) -> i32;
    pub fn jolt_eq_evals(ctx: *mut jolt_ctx, r: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_eq_evals_aligned_block(ctx: *mut jolt_ctx, r: *const jolt_fr_t, n: usize, start: usize, block: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_lt_evals(ctx: *mut jolt_ctx, r: *const jolt_fr_t, n: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_eq_plus_one_evals(ctx: *mut jolt_ctx, r: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, eq_out: *mut *mut jolt_table, eq_plus_one_out: *mut *mut jolt_table) -> i32;
    pub fn jolt_address_fold(ctx: *mut jolt_ctx, grid: *const jolt_table, weights: *const jolt_table, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_cycle_fold(ctx: *mut jolt_ctx, grid: *const jolt_table, weights: *const jolt_table, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_tile(ctx: *mut jolt_ctx, base: *const jolt_table, copies: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_replicate_stream_lsb(ctx: *mut jolt_ctx, base: *const jolt_table, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_rlc(ctx: *mut jolt_ctx, tables: *const *mut jolt_table, k: usize, scalars: *const jolt_fr_t, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_table_evaluate(ctx: *mut jolt_ctx, t: *const jolt_table, point: *const jolt_fr_t, n: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_table_sum(ctx: *mut jolt_ctx, t: *const jolt_table, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_member_create_expr(ctx: *mut jolt_ctx, tables: *const *mut jolt_table, desc: *const jolt_member_desc, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_member_create_lc(ctx: *mut jolt_ctx, tables: *const *mut jolt_table, desc: *const jolt_member_lc_desc, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_member_create_split_eq_product(ctx: *mut jolt_ctx, a: *mut jolt_table, b: *mut jolt_table, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_member_create_split_eq_product_borrowed(ctx: *mut jolt_ctx, a: *mut jolt_table, b: *mut jolt_table, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_member_create_split_eq_lc(ctx: *mut jolt_ctx, tables: *const *mut jolt_table, desc: *const jolt_member_lc_desc, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, shard_scale: *const jolt_fr_t, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_member_create_lc_small(ctx: *mut jolt_ctx, tables: *const *mut jolt_table, ints: *const *const jolt_ints, desc: *const jolt_member_lc_desc, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, shard_scale: *const jolt_fr_t, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_host_small_round_pair(desc: *const jolt_member_lc_desc, is_int: *const u8, int_pairs: *const u64, fr_pairs: *const jolt_fr_t, n_evals: u32, skip_one: i32, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_member_create_split_eq_uniform(ctx: *mut jolt_ctx, tables: *const *mut jolt_table, V: u32, F: u32, coeffs: *const jolt_fr_t, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, shard_scale: *const jolt_fr_t, flags: u32, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_member_create_split_eq_product_sharded(ctx: *mut jolt_ctx, a: *mut jolt_table, b: *mut jolt_table, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, shard_scale: *const jolt_fr_t, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_member_reset(m: *mut jolt_member) -> i32;
    pub fn jolt_member_set_scale(member: *mut jolt_member, scale: *const jolt_fr_t) -> i32;
    pub fn jolt_member_num_rounds(m: *const jolt_member, rounds: *mut usize) -> i32;
    pub fn jolt_member_degree(m: *const jolt_member, degree: *mut u32) -> i32;
    pub fn jolt_member_prove_round(m: *mut jolt_member, bind: *const jolt_fr_t, evals_out: *mut jolt_fr_t, n_evals: usize, aux_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_round_group_prove(ctx: *mut jolt_ctx, members: *const *mut jolt_member, n_members: usize, binds: *const *const jolt_fr_t, evals_out: *mut jolt_fr_t, evals_capacity: usize) -> i32;
    pub fn jolt_member_finish(m: *mut jolt_member, bind: *const jolt_fr_t) -> i32;
    pub fn jolt_round_group_finish(ctx: *mut jolt_ctx, members: *const *mut jolt_member, n_members: usize, binds: *const *const jolt_fr_t) -> i32;
    pub fn jolt_member_final_values(m: *mut jolt_member, out: *mut jolt_fr_t, k: usize) -> i32;
    pub fn jolt_member_input_claim(m: *mut jolt_member, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_member_destroy(m: *mut jolt_member) -> i32;
    pub fn jolt_srs_upload_g1(ctx: *mut jolt_ctx, bases: *const jolt_g1_t, n: usize, out: *mut *mut jolt_srs) -> i32;
    pub fn jolt_srs_setup_from_secret(ctx: *mut jolt_ctx, beta: *const jolt_fr_t, count: usize, g1: *const jolt_g1_t, out: *mut *mut jolt_srs) -> i32;
    pub fn jolt_srs_len(srs: *const jolt_srs, n: *mut usize) -> i32;
    pub fn jolt_srs_download(ctx: *mut jolt_ctx, srs: *const jolt_srs, offset: usize, n: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_srs_free(ctx: *mut jolt_ctx, srs: *mut jolt_srs) -> i32;
    pub fn jolt_srs_precompute_windows(ctx: *mut jolt_ctx, srs: *mut jolt_srs, window_bits: u32, min_terms: usize) -> i32;
    pub fn jolt_msm_profile_buckets(ctx: *mut jolt_ctx, enable: i32) -> i32;
    pub fn jolt_msm_profile_buckets_last(ctx: *mut jolt_ctx, ms: *mut f32, additions: *mut u64) -> i32;
    pub fn jolt_ctx_measure_mad_peak(ctx: *mut jolt_ctx, target_ms: f32, mads_per_s: *mut f64, timed_ms: *mut f32, launches: *mut u32) -> i32;
    pub fn jolt_msm_g1(ctx: *mut jolt_ctx, srs: *const jolt_srs, scalars: *const jolt_fr_t, n: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_msm_g1_table(ctx: *mut jolt_ctx, srs: *const jolt_srs, scalars: *const jolt_table, n: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_msm_g1_table_full_width(ctx: *mut jolt_ctx, srs: *const jolt_srs, scalars: *const jolt_table, n: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_msm_g1_tables_begin(ctx: *mut jolt_ctx, srs: *const jolt_srs, scalars: *const *const jolt_table, n: *const usize, count: usize, out: *mut *mut jolt_msm_pending) -> i32;
    pub fn jolt_msm_g1_tables_finish(ctx: *mut jolt_ctx, pending: *mut jolt_msm_pending, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_hyperkzg_fold(ctx: *mut jolt_ctx, evals: *const jolt_table, point: *const jolt_fr_t, ell: usize, levels_out: *mut *mut jolt_table) -> i32;
    pub fn jolt_hyperkzg_eval3(ctx: *mut jolt_ctx, levels: *const *mut jolt_table, ell: usize, u: *const jolt_fr_t, v_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_hyperkzg_rlc(ctx: *mut jolt_ctx, levels: *const *mut jolt_table, ell: usize, q: *const jolt_fr_t, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_hyperkzg_witness_poly(ctx: *mut jolt_ctx, f: *const jolt_table, u: *const jolt_fr_t, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_host_fr_mul(a: *const jolt_fr_t, b: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_fr_add(a: *const jolt_fr_t, b: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_fr_sub(a: *const jolt_fr_t, b: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_fr_inv(a: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_fr_from_u64(v: u64, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_fr_mul_shifted(a: *const jolt_fr_t, c: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_eq_evals(r: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_mul_limbs29(field: i32, a: *const jolt_fr_t, b: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_fq_limb_op(op: i32, a: *const jolt_fr_t, b: *const jolt_fr_t, c: *const jolt_fr_t, d: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_g1_sum_limb_form(points: *const u64, negate: *const u8, count: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_host_fx_digits(scalar: *const jolt_fr_t, window_bits: u32, keys_out: *mut u32, n_windows_out: *mut u32, buckets_out: *mut u32) -> i32;
    pub fn jolt_host_fx_segment_capacity(n: u64, window_bits: u32, segment: u32, capacity: *mut u32) -> i32;
    pub fn jolt_host_univariate_from_evals(evals: *const jolt_fr_t, n: usize, coeffs_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_univariate_evaluate(coeffs: *const jolt_fr_t, n: usize, x: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_gruen_poly_from_q(current_scalar: *const jolt_fr_t, point_i: *const jolt_fr_t, q_evals: *const jolt_fr_t, dq: usize, s0_plus_s1: *const jolt_fr_t, coeffs_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_gruen_poly_deg_3(current_scalar: *const jolt_fr_t, point_i: *const jolt_fr_t, q_constant: *const jolt_fr_t, q_quadratic: *const jolt_fr_t, s0_plus_s1: *const jolt_fr_t, coeffs_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_booleanity_address_round(linear: *const jolt_fr_t, squared: *const jolt_fr_t, n_polys: usize, stride: usize, len: usize, weights: *const jolt_fr_t, eq_address: *const jolt_fr_t, evals_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_booleanity_address_bind(linear: *mut jolt_fr_t, squared: *mut jolt_fr_t, n_polys: usize, stride: usize, len: usize, eq_address: *mut jolt_fr_t, challenge: *const jolt_fr_t) -> i32;
    pub fn jolt_host_hamming_weights(gamma: *const jolt_fr_t, r_address: *const jolt_fr_t, virtualization_points: *const jolt_fr_t, n_polys: usize, log_k: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_pair_tables_round(g: *const jolt_fr_t, w: *const jolt_fr_t, n_polys: usize, stride: usize, len: usize, evals_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_pair_tables_bind(g: *mut jolt_fr_t, w: *mut jolt_fr_t, n_polys: usize, stride: usize, len: usize, challenge: *const jolt_fr_t) -> i32;
    pub fn jolt_host_transcript_create(label: u64, out: *mut *mut jolt_host_transcript) -> i32;
    pub fn jolt_host_transcript_create_labelled(kind: i32, label: *const u8, label_len: usize, out: *mut *mut jolt_host_transcript) -> i32;
    pub fn jolt_host_transcript_append_fr(t: *mut jolt_host_transcript, values: *const jolt_fr_t, count: usize) -> i32;
    pub fn jolt_host_transcript_append_bytes(t: *mut jolt_host_transcript, bytes: *const u8, count: usize) -> i32;
    pub fn jolt_host_transcript_append_label(t: *mut jolt_host_transcript, label: *const c_char, with_count: i32, count: u64) -> i32;
    pub fn jolt_host_transcript_append_u64_word(t: *mut jolt_host_transcript, value: u64) -> i32;
    pub fn jolt_host_transcript_append_round_poly(t: *mut jolt_host_transcript, label: *const c_char, coefficients: *const jolt_fr_t, count: usize) -> i32;
    pub fn jolt_host_transcript_challenge(t: *mut jolt_host_transcript, full_width: i32, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_transcript_state(t: *const jolt_host_transcript, out32: *mut u8) -> i32;
    pub fn jolt_host_transcript_destroy(t: *mut jolt_host_transcript) -> i32;
    pub fn jolt_host_blake2b(r#in: *const u8, n: usize, outlen: usize, out: *mut u8) -> i32;
    pub fn jolt_host_keccak_f1600(state200: *mut u8) -> i32;
    pub fn jolt_host_g1_add(p: *const jolt_g1_t, q: *const jolt_g1_t, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_host_g1_eq(p: *const jolt_g1_t, q: *const jolt_g1_t, equal: *mut i32) -> i32;
    pub fn jolt_host_g1_is_on_curve(p: *const jolt_g1_t, on_curve: *mut i32) -> i32;
    pub fn jolt_host_g1_serialize_compressed(p: *const jolt_g1_t, out: *mut u8) -> i32;
    pub fn jolt_host_prove_batch(ctx: *mut jolt_ctx, members: *const *mut jolt_member, n_members: usize, input_claims: *const jolt_fr_t, coefficients: *const jolt_fr_t, offsets: *const usize, max_num_vars: usize, max_degree: usize, transcript_label: u64, challenge_mode: i32, use_round_group: i32, out_polys: *mut jolt_fr_t, out_challenges: *mut jolt_fr_t, out_member_claims: *mut jolt_fr_t, out_final_claim: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_batch_begin(ctx: *mut jolt_ctx, n_members: usize, input_claims: *const jolt_fr_t, coefficients: *const jolt_fr_t, rounds: *const usize, offsets: *const usize, kinds: *const i32, degrees: *const u32, split_eq_points: *const *const jolt_fr_t, split_eq_scales: *const jolt_fr_t, max_num_vars: usize, max_degree: usize, transcript_label: u64, challenge_mode: i32, out: *mut *mut jolt_batch) -> i32;
    pub fn jolt_host_batch_run(b: *mut jolt_batch, members: *const *mut jolt_member, n_rounds: usize, world: i32, gather: jolt_gather_fn, local_fn: jolt_local_round_fn, user: *mut c_void) -> i32;
    pub fn jolt_host_batch_set_transcript(b: *mut jolt_batch, r#fn: jolt_round_transcript_fn, user: *mut c_void) -> i32;
    pub fn jolt_host_batch_flush_binds(b: *mut jolt_batch, members: *const *mut jolt_member, binds_out: *mut jolt_fr_t, has_bind_out: *mut i32) -> i32;
    pub fn jolt_host_batch_split_eq_scalar(b: *const jolt_batch, member: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_batch_end(b: *mut jolt_batch, out_polys: *mut jolt_fr_t, out_challenges: *mut jolt_fr_t, out_member_claims: *mut jolt_fr_t, out_final_claim: *mut jolt_fr_t) -> i32;
    pub fn jolt_split_lt_create(ctx: *mut jolt_ctx, r_cycle: *const jolt_fr_t, n: usize, constant: *const jolt_fr_t, out: *mut *mut jolt_split_lt) -> i32;
    pub fn jolt_split_lt_bind(ctx: *mut jolt_ctx, s: *mut jolt_split_lt, r: *const jolt_fr_t) -> i32;
    pub fn jolt_split_lt_len(s: *const jolt_split_lt, len: *mut usize) -> i32;
    pub fn jolt_split_lt_to_dense(ctx: *mut jolt_ctx, s: *const jolt_split_lt, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_split_lt_final_value(ctx: *mut jolt_ctx, s: *const jolt_split_lt, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_split_lt_free(ctx: *mut jolt_ctx, s: *mut jolt_split_lt) -> i32;
    pub fn jolt_host_fr_wide_dot(a: *const jolt_fr_t, b: *const jolt_fr_t, n: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_table_dot(ctx: *mut jolt_ctx, a: *const jolt_table, b: *const jolt_table, deferred: i32, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_onehot_upload(ctx: *mut jolt_ctx, indices: *const u8, n_polys: usize, cycles: usize, k: u32, out: *mut *mut jolt_onehot) -> i32;
    pub fn jolt_onehot_upload16(ctx: *mut jolt_ctx, indices: *const u16, n_polys: usize, cycles: usize, k: u32, out: *mut *mut jolt_onehot) -> i32;
    pub fn jolt_onehot_download16(ctx: *mut jolt_ctx, source: *const jolt_onehot, out: *mut u16) -> i32;
    pub fn jolt_onehot_free(ctx: *mut jolt_ctx, source: *mut jolt_onehot) -> i32;
    pub fn jolt_onehot_materialize(ctx: *mut jolt_ctx, source: *const jolt_onehot, poly: usize, scale_table: *const jolt_table, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_onehot_pushforward(ctx: *mut jolt_ctx, source: *const jolt_onehot, weights: *const jolt_table, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_member_create_lazy_ra_uniform(ctx: *mut jolt_ctx, source: *const jolt_onehot, scale_tables: *const jolt_fr_t, V: u32, F: u32, coeffs: *const jolt_fr_t, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_member_create_lazy_ra_uniform_sharded(ctx: *mut jolt_ctx, source: *const jolt_onehot, scale_tables: *const jolt_fr_t, V: u32, F: u32, coeffs: *const jolt_fr_t, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, shard_scale: *const jolt_fr_t, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_rows_upload(ctx: *mut jolt_ctx, rows: *const c_void, n_rows: usize, row_bytes: usize, out: *mut *mut jolt_rows) -> i32;
    pub fn jolt_rows_upload_begin(ctx: *mut jolt_ctx, rows: *const c_void, n_rows: usize, row_bytes: usize, out: *mut *mut jolt_rows) -> i32;
    pub fn jolt_rows_upload_wait(ctx: *mut jolt_ctx, rows: *mut jolt_rows) -> i32;
    pub fn jolt_host_pinned_alloc(ctx: *mut jolt_ctx, bytes: usize, out: *mut *mut c_void) -> i32;
    pub fn jolt_host_pinned_free(ctx: *mut jolt_ctx, p: *mut c_void) -> i32;
    pub fn jolt_rows_free(ctx: *mut jolt_ctx, rows: *mut jolt_rows) -> i32;
    pub fn jolt_table_from_rows(ctx: *mut jolt_ctx, rows: *const jolt_rows, offset: usize, width: u32, is_signed: i32, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_ints_from_rows(ctx: *mut jolt_ctx, rows: *const jolt_rows, offset: usize, width: u32, is_signed: i32, out: *mut *mut jolt_ints) -> i32;
    pub fn jolt_ints_from_rows_many(ctx: *mut jolt_ctx, rows: *const jolt_rows, offsets: *const usize, widths: *const u32, is_signed: *const i32, n_fields: usize, out: *mut *mut jolt_ints) -> i32;
    pub fn jolt_onehot_from_rows(ctx: *mut jolt_ctx, rows: *const jolt_rows, offset: usize, width: u32, shifts: *const u32, n_polys: usize, log_k: u32, valid_offset: usize, out: *mut *mut jolt_onehot) -> i32;
    pub fn jolt_onehot_download(ctx: *mut jolt_ctx, source: *const jolt_onehot, out: *mut u8) -> i32;
    pub fn jolt_table_from_rows_window(ctx: *mut jolt_ctx, rows: *const jolt_rows, offset: usize, width: u32, is_signed: i32, lookahead: i32, cycles: usize, padding_value: i64, none_value: i64, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_onehot_from_rows_sentinel(ctx: *mut jolt_ctx, rows: *const jolt_rows, offset: usize, width: u32, shifts: *const u32, n_polys: usize, log_k: u32, cycles: usize, out: *mut *mut jolt_onehot) -> i32;
    pub fn jolt_member_create_lazy_booleanity(ctx: *mut jolt_ctx, source: *const jolt_onehot, scale_tables: *const jolt_fr_t, rho: *const jolt_fr_t, w: *const jolt_fr_t, n: usize, scale: *const jolt_fr_t, out: *mut *mut jolt_member) -> i32;
    pub fn jolt_ints_upload(ctx: *mut jolt_ctx, host: *const c_void, kind: i32, count: usize, out: *mut *mut jolt_ints) -> i32;
    pub fn jolt_ints_free(ctx: *mut jolt_ctx, values: *mut jolt_ints) -> i32;
    pub fn jolt_dory_commit_rows(ctx: *mut jolt_ctx, srs: *const jolt_srs, values: *const jolt_ints, row_width: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_dory_commit_onehot(ctx: *mut jolt_ctx, srs: *const jolt_srs, source: *const jolt_onehot, poly: usize, chunk_width: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_table_from_ints(ctx: *mut jolt_ctx, values: *const jolt_ints, offset: usize, len: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_grid_commit_onehot(ctx: *mut jolt_ctx, srs: *const jolt_srs, source: *const jolt_onehot, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_grid_joint_polynomial(ctx: *mut jolt_ctx, sources: *const *const jolt_onehot, n_sources: usize, onehot_scalars: *const jolt_fr_t, dense: *const *mut jolt_table, n_dense: usize, dense_scalars: *const jolt_fr_t, log_k: u32, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_read_raf_create(ctx: *mut jolt_ctx, lookup_index: *const u64, table_index: *const u8, raf_flag: *const u8, cycles: usize, n_tables: u32, out: *mut *mut jolt_read_raf) -> i32;
    pub fn jolt_read_raf_destroy(ctx: *mut jolt_ctx, rr: *mut jolt_read_raf) -> i32;
    pub fn jolt_read_raf_cycles(rr: *const jolt_read_raf, cycles: *mut usize, n_tables: *mut u32) -> i32;
    pub fn jolt_read_raf_phase_scan(ctx: *mut jolt_ctx, rr: *mut jolt_read_raf, u: *const jolt_table, suffix_len: u32, address_bits: u32, canonical: i32, suffix_offsets: *const u32, suffix_kinds: *const u8, raf_out: *mut jolt_fr_t, suffix_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_read_raf_condense(ctx: *mut jolt_ctx, rr: *mut jolt_read_raf, u: *mut jolt_table, v_table: *const jolt_fr_t, shift: u32) -> i32;
    pub fn jolt_read_raf_cycle_tables(ctx: *mut jolt_ctx, rr: *mut jolt_read_raf, table_values: *const jolt_fr_t, raf_interleaved: *const jolt_fr_t, raf_identity: *const jolt_fr_t, v_tables: *const jolt_fr_t, phases: u32, address_bits: u32, ra_count: u32, combined_out: *mut *mut jolt_table, ra_out: *mut *mut jolt_table) -> i32;
    pub fn jolt_host_suffix_mle(kind: u32, lo: u64, hi: u64, len: u32, out: *mut u64) -> i32;
    pub fn jolt_lookup_table_count() -> u32;
    pub fn jolt_lookup_prefix_count() -> u32;
    pub fn jolt_lookup_table_suffixes(kind: u32, kinds_out: *mut u8, n_out: *mut u32) -> i32;
    pub fn jolt_lookup_table_prefixes(kind: u32, prefixes_out: *mut u8, n_out: *mut u32) -> i32;
    pub fn jolt_lookup_suffix_layout(offsets_out: *mut u32, kinds_out: *mut u8) -> i32;
    pub fn jolt_host_lookup_prefix_default_checkpoints(out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_lookup_prefix_evaluate(prefix: u32, checkpoints: *const jolt_fr_t, b: u32, b_len: u32, suffix_len: u32, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_lookup_prefix_table(prefix: u32, checkpoints: *const jolt_fr_t, b_len: u32, suffix_len: u32, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_lookup_table_combine(kind: u32, prefixes: *const jolt_fr_t, suffixes: *const jolt_fr_t, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_read_raf_address_create(gamma: *const jolt_fr_t, table_present: *const u8, canonical: i32, out: *mut *mut jolt_read_raf_address) -> i32;
    pub fn jolt_host_read_raf_address_destroy(h: *mut jolt_read_raf_address) -> i32;
    pub fn jolt_host_read_raf_address_init_phase(h: *mut jolt_read_raf_address, phase: u32, raf_sums: *const jolt_fr_t, suffix_sums: *const jolt_fr_t) -> i32;
    pub fn jolt_host_read_raf_address_message(h: *mut jolt_read_raf_address, previous_claim: *const jolt_fr_t, evals_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_read_raf_address_bind(h: *mut jolt_read_raf_address, challenge: *const jolt_fr_t, phase_done: *mut i32) -> i32;
    pub fn jolt_host_read_raf_address_bind_message(h: *mut jolt_read_raf_address, challenge: *const jolt_fr_t, previous_claim: *const jolt_fr_t, evals_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_read_raf_address_prove_phase(h: *mut jolt_read_raf_address, claim: *mut jolt_fr_t, r#fn: jolt_round_transcript_fn, user: *mut c_void, test_transcript: *mut jolt_host_transcript, coeffs_out: *mut jolt_fr_t, challenges_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_read_raf_address_v_table(h: *const jolt_read_raf_address, phase: u32, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_read_raf_address_finish(h: *const jolt_read_raf_address, table_values: *mut jolt_fr_t, raf_interleaved: *mut jolt_fr_t, raf_identity: *mut jolt_fr_t) -> i32;
    pub fn jolt_r1cs_uniskip_sums(ctx: *mut jolt_ctx, inputs: *const *mut jolt_table, n_inputs: usize, eq: *const jolt_table, a_weights: *const jolt_fr_t, b_weights: *const jolt_fr_t, n_nodes: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_r1cs_materialize(ctx: *mut jolt_ctx, inputs: *const *mut jolt_table, n_inputs: usize, a_weights: *const jolt_fr_t, b_weights: *const jolt_fr_t, az_out: *mut *mut jolt_table, bz_out: *mut *mut jolt_table) -> i32;
    pub fn jolt_tables_evaluate(ctx: *mut jolt_ctx, tables: *const *mut jolt_table, k: usize, point: *const jolt_fr_t, n: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_r1cs_uniskip_sums_small(ctx: *mut jolt_ctx, inputs: *const *const jolt_ints, n_inputs: usize, eq: *const jolt_table, n_streams: u32, a_weights: *const i64, b_weights: *const i64, n_nodes: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_r1cs_materialize_small(ctx: *mut jolt_ctx, inputs: *const *const jolt_ints, n_inputs: usize, n_streams: u32, a_weights: *const jolt_fr_t, b_weights: *const jolt_fr_t, az_out: *mut *mut jolt_table, bz_out: *mut *mut jolt_table) -> i32;
    pub fn jolt_ints_evaluate(ctx: *mut jolt_ctx, columns: *const *const jolt_ints, k: usize, point: *const jolt_fr_t, n: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_small_scalar_dot(values: *const jolt_fr_t, scalars: *const u64, n: usize, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_rw_matrix_create(ctx: *mut jolt_ctx, addresses: *const u64, pre_values: *const u64, post_values: *const u64, cycles: usize, inc: *const jolt_table, val_init: *const jolt_table, tau_low: *const jolt_fr_t, gamma: *const jolt_fr_t, out: *mut *mut jolt_rw_matrix) -> i32;
    pub fn jolt_rw_matrix_create_resident(ctx: *mut jolt_ctx, addresses: *const jolt_ints, pre_values: *const jolt_ints, post_values: *const jolt_ints, inc: *const jolt_table, val_init: *const jolt_table, tau_low: *const jolt_fr_t, gamma: *const jolt_fr_t, out: *mut *mut jolt_rw_matrix) -> i32;
    pub fn jolt_registers_rw_create(ctx: *mut jolt_ctx, regs: *const jolt_onehot, rs1_val: *const jolt_ints, rs2_val: *const jolt_ints, rd_pre: *const jolt_ints, rd_post: *const jolt_ints, inc: *const jolt_table, r_cycle: *const jolt_fr_t, gamma: *const jolt_fr_t, out: *mut *mut jolt_rw_matrix) -> i32;
    pub fn jolt_registers_rw_prove_round(m: *mut jolt_rw_matrix, bind: *const jolt_fr_t, evals_out: *mut jolt_fr_t, aux_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_registers_rw_final_values(m: *mut jolt_rw_matrix, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_registers_rw_download(m: *mut jolt_rw_matrix, rows: *mut u64, cols: *mut u64, val: *mut jolt_fr_t, ra: *mut jolt_fr_t, wa: *mut jolt_fr_t, prev: *mut u64, next: *mut u64) -> i32;
    pub fn jolt_rw_matrix_prove_round(m: *mut jolt_rw_matrix, bind: *const jolt_fr_t, evals_out: *mut jolt_fr_t, aux_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_rw_matrix_finish(m: *mut jolt_rw_matrix, bind: *const jolt_fr_t) -> i32;
    pub fn jolt_rw_matrix_final_values(m: *mut jolt_rw_matrix, out: *mut jolt_fr_t) -> i32;
    pub fn jolt_key_index_create(ctx: *mut jolt_ctx, keys: *const jolt_ints, k: u64, out: *mut *mut jolt_key_index) -> i32;
    pub fn jolt_key_index_size(index: *const jolt_key_index, cycles: *mut usize, k: *mut u64, items: *mut u32) -> i32;
    pub fn jolt_key_index_pushforward(ctx: *mut jolt_ctx, index: *const jolt_key_index, weights: *const *mut jolt_table, n_weights: usize, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_key_index_last_value(ctx: *mut jolt_ctx, index: *const jolt_key_index, values: *const jolt_ints, init: *const jolt_table, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_key_index_destroy(ctx: *mut jolt_ctx, index: *mut jolt_key_index) -> i32;
    pub fn jolt_rw_matrix_hold_row(m: *mut jolt_rw_matrix) -> i32;
    pub fn jolt_rw_matrix_bind(m: *mut jolt_rw_matrix, bind: *const jolt_fr_t) -> i32;
    pub fn jolt_rw_matrix_export_row(m: *mut jolt_rw_matrix, cap: usize, cols: *mut u64, prev: *mut u64, next: *mut u64, val: *mut jolt_fr_t, ra: *mut jolt_fr_t, wa: *mut jolt_fr_t, inc_out: *mut jolt_fr_t, scalar_out: *mut jolt_fr_t, n_out: *mut usize) -> i32;
    pub fn jolt_rw_matrix_create_merged(ctx: *mut jolt_ctx, registers: i32, log_rows: usize, log_k: usize, n: usize, rows: *const u64, cols: *const u64, prev: *const u64, next: *const u64, val: *const jolt_fr_t, ra: *const jolt_fr_t, wa: *const jolt_fr_t, inc: *const jolt_fr_t, val_init: *const jolt_table, w: *const jolt_fr_t, scalar: *const jolt_fr_t, gamma: *const jolt_fr_t, out: *mut *mut jolt_rw_matrix) -> i32;
    pub fn jolt_rw_matrix_len(m: *const jolt_rw_matrix, entries: *mut usize) -> i32;
    pub fn jolt_rw_matrix_download(m: *mut jolt_rw_matrix, rows: *mut u64, cols: *mut u64, val: *mut jolt_fr_t, ra: *mut jolt_fr_t, prev: *mut jolt_fr_t, next: *mut jolt_fr_t) -> i32;
    pub fn jolt_rw_matrix_destroy(m: *mut jolt_rw_matrix) -> i32;
    pub fn jolt_comm_unique_id(rccl_path: *const c_char, out: *mut u8) -> i32;
    pub fn jolt_comm_create(ctx: *mut jolt_ctx, rccl_path: *const c_char, unique_id: *const u8, rank: i32, world: i32, out: *mut *mut jolt_comm) -> i32;
    pub fn jolt_comm_destroy(comm: *mut jolt_comm) -> i32;
    pub fn jolt_comm_world(comm: *const jolt_comm, rank: *mut i32, world: *mut i32) -> i32;
    pub fn jolt_comm_all_gather_host(comm: *mut jolt_comm, local: *const c_void, bytes: usize, gathered: *mut c_void) -> i32;
    pub fn jolt_comm_all_gather_device(comm: *mut jolt_comm, d_local: *const c_void, bytes: usize, d_gathered: *mut c_void) -> i32;
    pub fn jolt_comm_all_gather_table(comm: *mut jolt_comm, local: *const jolt_table, n: usize, gathered: *mut jolt_table) -> i32;
    pub fn jolt_comm_gather_round_sums(user: *mut c_void, local: *const jolt_fr_t, count: usize, gathered: *mut jolt_fr_t) -> i32;
    pub fn jolt_shm_create(name: *const c_char, rank: i32, world: i32, max_bytes: usize, out: *mut *mut jolt_shm) -> i32;
    pub fn jolt_shm_create_nonce(name: *const c_char, nonce: u64, rank: i32, world: i32, max_bytes: usize, out: *mut *mut jolt_shm) -> i32;
    pub fn jolt_shm_destroy(shm: *mut jolt_shm) -> i32;
    pub fn jolt_shm_all_gather(shm: *mut jolt_shm, local: *const c_void, bytes: usize, gathered: *mut c_void) -> i32;
    pub fn jolt_shm_gather_round_sums(user: *mut c_void, local: *const jolt_fr_t, count: usize, gathered: *mut jolt_fr_t) -> i32;
    pub fn jolt_round_group_pack_tables(ctx: *mut jolt_ctx, members: *const *mut jolt_member, n_members: usize, entries: usize, dst: *mut jolt_table) -> i32;
    pub fn jolt_tail_interleave(ctx: *mut jolt_ctx, gathered: *const jolt_table, world: usize, n_tables: usize, entries: usize, dst: *mut jolt_table) -> i32;
    pub fn jolt_round_group_final_values(ctx: *mut jolt_ctx, members: *const *mut jolt_member, n_members: usize, out: *mut jolt_fr_t, capacity: usize) -> i32;
    pub fn jolt_host_hyperkzg_commit(ctx: *mut jolt_ctx, srs: *const jolt_srs, evals: *const jolt_table, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_host_hyperkzg_open(ctx: *mut jolt_ctx, srs: *const jolt_srs, evals: *const jolt_table, point: *const jolt_fr_t, ell: usize, transcript_label: u64, com: *mut jolt_g1_t, w: *mut jolt_g1_t, v: *mut jolt_fr_t, challenges_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_hyperkzg_open_with_transcript(ctx: *mut jolt_ctx, srs: *const jolt_srs, evals: *const jolt_table, point: *const jolt_fr_t, ell: usize, r#fn: jolt_open_transcript_fn, user: *mut c_void, com: *mut jolt_g1_t, w: *mut jolt_g1_t, v: *mut jolt_fr_t, challenges_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_hyperkzg_open_with_levels(ctx: *mut jolt_ctx, srs: *const jolt_srs, evals: *const jolt_table, point: *const jolt_fr_t, ell: usize, transcript_label: u64, r#fn: jolt_open_transcript_fn, user: *mut c_void, known_levels: *const jolt_g1_t, n_known: usize, com: *mut jolt_g1_t, w: *mut jolt_g1_t, v: *mut jolt_fr_t, challenges_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_msm_g1_table_range(ctx: *mut jolt_ctx, srs: *const jolt_srs, base_offset: usize, scalars: *const jolt_table, scalar_offset: usize, n: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_msm_g1_window(ctx: *mut jolt_ctx, srs: *const jolt_srs, base_offset: usize, kind: i32, host: *const c_void, n: usize, acc: *const jolt_g1_t, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_grid_commit_onehot_range(ctx: *mut jolt_ctx, srs: *const jolt_srs, source: *const jolt_onehot, cycle_lo: usize, cycle_hi: usize, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_grid_commit_onehot_classes(ctx: *mut jolt_ctx, srs: *const jolt_srs, source: *const jolt_onehot, shift: u32, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_host_hyperkzg_open_sharded(ctx: *mut jolt_ctx, srs: *const jolt_srs, evals: *const jolt_table, point: *const jolt_fr_t, ell: usize, transcript_label: u64, rank: i32, world: i32, gather: jolt_gather_fn, user: *mut c_void, com: *mut jolt_g1_t, w: *mut jolt_g1_t, v: *mut jolt_fr_t, challenges_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_owned_terms(n: usize, block: usize, rank: i32, world: i32, out: *mut usize) -> i32;
    pub fn jolt_srs_setup_from_secret_blocks(ctx: *mut jolt_ctx, beta: *const jolt_fr_t, count_global: usize, g1: *const jolt_g1_t, block: usize, rank: i32, world: i32, out: *mut *mut jolt_srs) -> i32;
    pub fn jolt_msm_g1_table_blocks(ctx: *mut jolt_ctx, srs: *const jolt_srs, scalars: *const jolt_table, n: usize, block: usize, rank: i32, world: i32, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_host_hyperkzg_open_sharded_blocks(ctx: *mut jolt_ctx, srs: *const jolt_srs, evals: *const jolt_table, point: *const jolt_fr_t, ell: usize, transcript_label: u64, rank: i32, world: i32, block: usize, gather: jolt_gather_fn, user: *mut c_void, com: *mut jolt_g1_t, w: *mut jolt_g1_t, v: *mut jolt_fr_t, challenges_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_subtree_owned_terms(n: usize, rank: i32, world: i32, out: *mut usize) -> i32;
    pub fn jolt_host_subtree_term_index(slot: usize, rank: i32, world: i32, out: *mut usize) -> i32;
    pub fn jolt_srs_setup_from_secret_subtree(ctx: *mut jolt_ctx, beta: *const jolt_fr_t, count_global: usize, g1: *const jolt_g1_t, rank: i32, world: i32, out: *mut *mut jolt_srs) -> i32;
    pub fn jolt_msm_g1_table_subtree(ctx: *mut jolt_ctx, srs: *const jolt_srs, scalars: *const jolt_table, n: usize, rank: i32, world: i32, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_grid_joint_polynomial_subtree(ctx: *mut jolt_ctx, sources: *const *const jolt_onehot, n_sources: usize, onehot_scalars: *const jolt_fr_t, dense: *const *mut jolt_table, n_dense: usize, dense_scalars: *const jolt_fr_t, log_k: u32, rank: i32, world: i32, out: *mut *mut jolt_table) -> i32;
    pub fn jolt_host_hyperkzg_open_subtree(ctx: *mut jolt_ctx, srs: *const jolt_srs, evals: *const jolt_table, point: *const jolt_fr_t, ell: usize, transcript_label: u64, rank: i32, world: i32, gather: jolt_gather_fn, user: *mut c_void, com: *mut jolt_g1_t, w: *mut jolt_g1_t, v: *mut jolt_fr_t, challenges_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_grid_hint_begin(ctx: *mut jolt_ctx, srs: *const jolt_srs, sources: *const *const jolt_onehot, n_sources: usize, levels: u32, background: i32, out: *mut *mut jolt_grid_hint) -> i32;
    pub fn jolt_grid_hint_wait(ctx: *mut jolt_ctx, hint: *mut jolt_grid_hint) -> i32;
    pub fn jolt_grid_hint_download(ctx: *mut jolt_ctx, hint: *mut jolt_grid_hint, level: u32, out: *mut jolt_g1_t) -> i32;
    pub fn jolt_grid_hint_free(ctx: *mut jolt_ctx, hint: *mut jolt_grid_hint) -> i32;
    pub fn jolt_host_hyperkzg_open_grid(ctx: *mut jolt_ctx, srs: *const jolt_srs, evals: *const jolt_table, point: *const jolt_fr_t, ell: usize, transcript_label: u64, r#fn: jolt_open_transcript_fn, user: *mut c_void, hint: *const jolt_grid_hint, levels: u32, onehot_scalars: *const jolt_fr_t, dense: *const *mut jolt_table, n_dense: usize, dense_scalars: *const jolt_fr_t, com: *mut jolt_g1_t, w: *mut jolt_g1_t, v: *mut jolt_fr_t, challenges_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_stage_op_num_rounds(op: *const jolt_stage_op, rounds: *mut usize) -> i32;
    pub fn jolt_stage_op_degree(op: *const jolt_stage_op, degree: *mut usize) -> i32;
    pub fn jolt_stage_op_input_claim(op: *mut jolt_stage_op, claim: *mut jolt_fr_t) -> i32;
    pub fn jolt_stage_op_prove_round(op: *mut jolt_stage_op, bind: *const jolt_fr_t, round: usize, previous_claim: *const jolt_fr_t, coeffs_out: *mut jolt_fr_t, cap: usize, n_coeffs: *mut usize) -> i32;
    pub fn jolt_stage_op_finish_rounds(op: *mut jolt_stage_op, bind: *const jolt_fr_t) -> i32;
    pub fn jolt_stage_op_output_claims(op: *mut jolt_stage_op, out: *mut jolt_fr_t, cap: usize, n: *mut usize) -> i32;
    pub fn jolt_stage_op_kept(op: *const jolt_stage_op, key: *const c_char, out: *mut jolt_fr_t, cap: usize, n: *mut usize) -> i32;
    pub fn jolt_stage_op_window(parent: *mut jolt_stage_op, first: usize, n: usize, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_op_destroy(op: *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_spartan_uniskip_sums(ctx: *mut jolt_ctx, cols: *const *const jolt_ints, n_cols: usize, n_streams: u32, tau: *const jolt_fr_t, n_tau: usize, a_weights: *const i64, b_weights: *const i64, n_nodes: usize, sums_out: *mut jolt_fr_t) -> i32;
    pub fn jolt_stage_spartan_remainder_create(ctx: *mut jolt_ctx, cols: *const *const jolt_ints, n_cols: usize, n_streams: u32, a_weights: *const jolt_fr_t, b_weights: *const jolt_fr_t, tau: *const jolt_fr_t, n_tau: usize, scale: *const jolt_fr_t, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_ram_read_write_create(ctx: *mut jolt_ctx, addresses: *const jolt_ints, pre_values: *const jolt_ints, post_values: *const jolt_ints, inc: *const jolt_ints, val_init: *const jolt_ints, tau_low: *const jolt_fr_t, gamma: *const jolt_fr_t, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_registers_read_write_create(ctx: *mut jolt_ctx, regs: *const jolt_onehot, rs1_val: *const jolt_ints, rs2_val: *const jolt_ints, rd_pre: *const jolt_ints, rd_post: *const jolt_ints, inc: *const jolt_ints, r_cycle: *const jolt_fr_t, gamma: *const jolt_fr_t, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_booleanity_address_create(ctx: *mut jolt_ctx, cols: *const jolt_onehot, reference_cycle: *const jolt_fr_t, n_cycle: usize, reference_address: *const jolt_fr_t, gamma: *const jolt_fr_t, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_booleanity_cycle_create(ctx: *mut jolt_ctx, cols: *const jolt_onehot, r_address: *const jolt_fr_t, reference_address: *const jolt_fr_t, reference_cycle: *const jolt_fr_t, n_cycle: usize, gamma: *const jolt_fr_t, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_hamming_weight_create(ctx: *mut jolt_ctx, cols: *const jolt_onehot, r_cycle: *const jolt_fr_t, n_cycle: usize, r_address: *const jolt_fr_t, virtualization_points: *const jolt_fr_t, gamma: *const jolt_fr_t, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_instruction_read_raf_create(ctx: *mut jolt_ctx, rows: *mut jolt_read_raf, claim_columns: *const jolt_onehot, r_reduction: *const jolt_fr_t, n_vars: usize, gamma: *const jolt_fr_t, table_present: *const u8, ra_count: u32, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_bytecode_read_raf_address_create(ctx: *mut jolt_ctx, pc_index: *const jolt_key_index, stage_points: *const jolt_fr_t, n_vars: usize, stage_values: *const jolt_fr_t, gamma: *const jolt_fr_t, first_pc: u64, entry_index: u64, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_bytecode_read_raf_cycle_create(ctx: *mut jolt_ctx, address: *mut jolt_stage_op, pc_chunks: *const jolt_onehot, chunk_bits: u32, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_ram_raf_evaluation_create(ctx: *mut jolt_ctx, ram_index: *const jolt_key_index, tau_low: *const jolt_fr_t, n_vars: usize, lowest_address: u64, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_ram_output_check_create(ctx: *mut jolt_ctx, ram_index: *const jolt_key_index, post_values: *const jolt_ints, val_init: *const u64, val_io: *const u64, io_lo: u64, io_len: u64, r_address: *const jolt_fr_t, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_stage_host_expr_create(tables: *const *const jolt_fr_t, len: usize, desc: *const jolt_member_desc, out: *mut *mut jolt_stage_op) -> i32;
    pub fn jolt_host_prove_batch_ops(ctx: *mut jolt_ctx, ops: *const *mut jolt_stage_op, n_ops: usize, input_claims: *const jolt_fr_t, coefficients: *const jolt_fr_t, offsets: *const usize, max_num_vars: usize, max_degree: usize, transcript_label: u64, challenge_mode: i32, out_polys: *mut jolt_fr_t, out_challenges: *mut jolt_fr_t, out_member_claims: *mut jolt_fr_t, out_final_claim: *mut jolt_fr_t) -> i32;
    pub fn jolt_host_stage_op_prove_alone(op: *mut jolt_stage_op, transcript: *mut jolt_host_transcript, claim: *mut jolt_fr_t, coeffs_out: *mut jolt_fr_t, stride: usize, n_coeffs_out: *mut u32, challenges_out: *mut jolt_fr_t) -> i32;
}
