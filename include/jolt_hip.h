/*
 * include/jolt_hip.h -- C ABI of the MI355X-native Jolt prover hot path (libjolt_hip.so).
 *
 * This is the drop-in boundary: the entry points below are what a thin Rust FFI crate (`jolt-kernels-hip`,
 * sketched in INTEGRATION.md) binds in order to implement the reference's own trait surface for this path --
 * `ProveRounds`/`SumcheckKernel`/`PrepareKernel` (jolt-sumcheck, jolt-kernels), `JoltGroup::msm` (jolt-crypto) and
 * `CommitmentScheme::{commit,open}` for HyperKZG (jolt-openings / jolt-hyperkzg).  Every function cites the
 * reference interface it replaces (paths relative to the a16z/jolt checkout).
 *
 * Conventions:
 *  - plain pointers and sizes only; every function returns an int32 status and never unwinds/aborts;
 *  - jolt_fr_t  = 4 x u64 little-endian Montgomery limbs, canonical (< r)   == jolt_field::Fr::inner_limbs()
 *                 (crates/jolt-field/src/bn254/mod.rs:33-43);
 *  - jolt_g1_t  = Jacobian (x, y, z) of Montgomery Fq limbs, identity <=> z == 0 == ark_bn254::G1Projective, which
 *                 jolt_crypto::Bn254G1 wraps #[repr(transparent)] (crates/jolt-crypto/src/ec/bn254/mod.rs:17-24);
 *  - handles are opaque and owned by the context that created them; a kernel/member owns its tables
 *    (Box<dyn SumcheckKernel> semantics) and frees them on destroy;
 *  - all work is enqueued on the context's HIP stream; functions that return field elements or points to host
 *    memory synchronise that stream before returning (the per-round Fiat-Shamir sync point,
 *    crates/jolt-sumcheck/src/prover.rs:326);
 *  - there is NO CPU fallback: without a usable gfx950 device jolt_ctx_create fails with JOLT_ERR_NO_DEVICE.
 */
#ifndef JOLT_HIP_H
#define JOLT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JOLT_HIP_ABI_VERSION 1

typedef struct { uint64_t l[4]; } jolt_fr_t;
typedef struct { uint64_t l[4]; } jolt_fq_t;
typedef struct { jolt_fq_t x, y, z; } jolt_g1_t;

typedef struct jolt_ctx jolt_ctx;       /* device + stream + scratch                                  */
typedef struct jolt_table jolt_table;   /* device-resident dense table of Fr (Polynomial<Fr>)         */
typedef struct jolt_member jolt_member; /* one sumcheck batch member (Box<dyn SumcheckKernel>)        */
typedef struct jolt_srs jolt_srs;       /* device-resident affine SRS prefix (HyperKZGProverSetup.g1_powers) */

/* Status codes.  Mapping to the reference's error types (crates/jolt-kernels/src/error.rs:11-90,
 * crates/jolt-sumcheck/src/error.rs, crates/jolt-hyperkzg/src/error.rs) is given per value. */
enum {
    JOLT_OK = 0,
    JOLT_ERR_INVALID_ARG = 1,     /* KernelError::InvariantViolation (caller bug)                       */
    JOLT_ERR_NO_DEVICE = 2,       /* KernelError::Unsupported -> caller falls back to another backend   */
    JOLT_ERR_OOM = 3,             /* KernelError::Unsupported                                           */
    JOLT_ERR_HIP = 4,             /* KernelError::InvariantViolation, message via jolt_last_error       */
    JOLT_ERR_SIZE_MISMATCH = 5,   /* KernelError::TableSizeMismatch / msm length-mismatch panic         */
    JOLT_ERR_UNSUPPORTED = 6,     /* KernelError::Unsupported (descriptor beyond compiled limits)       */
    JOLT_ERR_NOT_FULLY_BOUND = 7, /* SumcheckKernelError::NotFullyBound                                 */
    JOLT_ERR_ROUND_CHECK = 8,     /* SumcheckError::RoundCheckFailed                                    */
    JOLT_ERR_SRS_TOO_SMALL = 9,   /* HyperKZGError::SrsTooSmall                                         */
    JOLT_ERR_EMPTY_POINT = 10,    /* HyperKZGError::EmptyPoint                                          */
    JOLT_ERR_NOT_INVERTIBLE = 11  /* gruen_poly_deg_3 `expect` (split_eq.rs:403-405)                    */
};

/* BindingOrder (crates/jolt-poly/src/binding.rs) */
enum { JOLT_ORDER_LOW_TO_HIGH = 0, JOLT_ORDER_HIGH_TO_LOW = 1 };

const char *jolt_status_string(int32_t status);
int32_t jolt_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Context.  One per GPU per process (one process per GPU; ranks shard the hypercube).
 * `stream` = an existing hipStream_t to enqueue on (e.g. the caller's), or NULL to create a private one.
 * ---------------------------------------------------------------------------------------------------------- */
int32_t jolt_ctx_create(int32_t device_id, void *stream, jolt_ctx **out);
int32_t jolt_ctx_destroy(jolt_ctx *ctx);
int32_t jolt_ctx_synchronize(jolt_ctx *ctx);
/* ... without the background stream (jolt_grid_hint_begin's class sums may outlive the leg that began them): what a caller timing a leg waits for */
int32_t jolt_ctx_synchronize_foreground(jolt_ctx *ctx);
/* Select the context's device on the CALLING host thread (the runtime's current device is per thread).  Contexts are single-threaded objects; two contexts on one device may be
 * driven from two threads at once -- the stage operators of one protocol stage beside that stage's batched sumcheck, as the reference batches them (crates/jolt-prover/src/stages). */
int32_t jolt_ctx_bind_thread(jolt_ctx *ctx);
const char *jolt_last_error(const jolt_ctx *ctx);
/* Device memory of tables, members and temporaries comes from a per-context pool (a proof builds and drops dozens of T-sized
 * derived tables -- the Vec<Fr> allocations of EqPolynomial::evals & co. in the reference; hipMalloc / hipFree cost 0.1-1 ms each
 * and synchronise the device).  _trim returns the cached blocks to the runtime (synchronises); _memory_stats reports the bytes held
 * by live handles, cached by the pool, and the high-water mark of both (any pointer may be NULL). */
int32_t jolt_ctx_trim(jolt_ctx *ctx);
int32_t jolt_ctx_memory_stats(const jolt_ctx *ctx, size_t *live_bytes, size_t *cached_bytes, size_t *peak_bytes);
/* ... and what the context holds outside the pool: the grow-only workspaces of its MSM lanes and of the batch of short MSMs (bytes). */
int32_t jolt_ctx_workspace_stats(const jolt_ctx *ctx, size_t *msm_lane_bytes, size_t *msm_batch_bytes);
/* Device-event timing of everything enqueued between begin and end on the context's stream (milliseconds). */
int32_t jolt_timer_begin(jolt_ctx *ctx);
int32_t jolt_timer_end(jolt_ctx *ctx, float *elapsed_ms);

/* ------------------------------------------------------------------------------------------------------------
 * Tables: Polynomial<Fr> resident in HBM (crates/jolt-poly/src/dense.rs:36-39).
 * Replaces the host Vec<Fr> that `witness.oracle_table(id)` / `dense_view` hand to kernels
 * (crates/jolt-witness/src/backend/mod.rs:50-53, crates/jolt-kernels/src/reference/views.rs:20-33).
 * ---------------------------------------------------------------------------------------------------------- */
int32_t jolt_table_upload(jolt_ctx *ctx, const jolt_fr_t *host, size_t len, jolt_table **out);
int32_t jolt_table_from_device(jolt_ctx *ctx, const void *device_ptr, size_t len, jolt_table **out); /* D2D copy */
int32_t jolt_table_alloc(jolt_ctx *ctx, size_t len, jolt_table **out);                               /* zeroed   */
int32_t jolt_table_clone(jolt_ctx *ctx, const jolt_table *src, jolt_table **out);
/* Small-scalar promotion on device: Ring::from_u64 / from_i64 per entry (crates/jolt-field/src/bn254/mod.rs:265-292),
 * the device twin of Polynomial<T>::bind_to_field's `From<T>` (dense.rs:129-142). */
int32_t jolt_table_from_u64(jolt_ctx *ctx, const uint64_t *host, size_t len, jolt_table **out);
int32_t jolt_table_from_i64(jolt_ctx *ctx, const int64_t *host, size_t len, jolt_table **out);
int32_t jolt_table_download(jolt_ctx *ctx, const jolt_table *t, size_t offset, size_t len, jolt_fr_t *host);
int32_t jolt_table_len(const jolt_table *t, size_t *len);
int32_t jolt_table_device_ptr(const jolt_table *t, void **device_ptr); /* current evaluations, len*32 bytes */
int32_t jolt_table_free(jolt_ctx *ctx, jolt_table *t);
/* A read-only view of entries [offset, offset+len) of `parent` (which must outlive it); overwrite a range from the host. */
int32_t jolt_table_slice(jolt_ctx *ctx, const jolt_table *parent, size_t offset, size_t len, jolt_table **out);
int32_t jolt_table_write(jolt_ctx *ctx, jolt_table *t, size_t offset, const jolt_fr_t *host, size_t len);

/* Polynomial::bind_with_order on k tables in ONE launch (dense.rs:178-263; optimized/support.rs:224-231 bind_all):
 *   LowToHigh: t[y] <- t[2y] + r*(t[2y+1]-t[2y]);   HighToLow: t[i] <- t[i] + r*(t[i+half]-t[i]); len halves. */
int32_t jolt_bind(jolt_ctx *ctx, jolt_table *const *tables, size_t k, const jolt_fr_t *r, int32_t order);

/* EqPolynomial::evals(r, scaling_factor) (crates/jolt-poly/src/eq.rs:221-231): 2^n entries, big-endian index
 * (r[0] pairs the MSB); scale may be NULL (= one). */
int32_t jolt_eq_evals(jolt_ctx *ctx, const jolt_fr_t *r, size_t n, const jolt_fr_t *scale, jolt_table **out);
/* EqPolynomial::evals_for_aligned_block (eq.rs:238-263): entries [start, start+block) only -- the per-GPU shard. */
int32_t jolt_eq_evals_aligned_block(jolt_ctx *ctx, const jolt_fr_t *r, size_t n, size_t start, size_t block,
                                    jolt_table **out);
/* LtPolynomial::evaluations (crates/jolt-poly/src/lt.rs:115-117,144-156) */
int32_t jolt_lt_evals(jolt_ctx *ctx, const jolt_fr_t *r, size_t n, jolt_table **out);
/* EqPlusOnePolynomial::evals (crates/jolt-poly/src/eq_plus_one.rs:71-130): (eq, eq+1) tables */
int32_t jolt_eq_plus_one_evals(jolt_ctx *ctx, const jolt_fr_t *r, size_t n, const jolt_fr_t *scale, jolt_table **eq_out,
                               jolt_table **eq_plus_one_out);
/* Derived-table builders of the reference tier (crates/jolt-kernels/src/reference/views.rs:35-138):
 *   address_fold: out[j] = sum_k w[k] * grid[(k << log_t) | j]      (grid address-major K x T, w has K entries)
 *   cycle_fold:   out[k] = sum_j w[j] * grid[(k << log_t) | j]      (w has T entries)
 *   tile: `copies` concatenated copies of base;  replicate_stream_lsb: out[(t << 1) | s] = base[t] */
int32_t jolt_address_fold(jolt_ctx *ctx, const jolt_table *grid, const jolt_table *weights, jolt_table **out);
int32_t jolt_cycle_fold(jolt_ctx *ctx, const jolt_table *grid, const jolt_table *weights, jolt_table **out);
int32_t jolt_tile(jolt_ctx *ctx, const jolt_table *base, size_t copies, jolt_table **out);
int32_t jolt_replicate_stream_lsb(jolt_ctx *ctx, const jolt_table *base, jolt_table **out);
/* Joint polynomial sum_i scalars[i] * f_i of a homomorphic batch opening: RlcSource::to_dense
 * (crates/jolt-poly/src/multilinear.rs:159-170,358-464) as consumed by HomomorphicBatch::prove_batch
 * (crates/jolt-openings/src/schemes.rs:487-524); k <= 40 tables of equal length. */
int32_t jolt_rlc(jolt_ctx *ctx, jolt_table *const *tables, size_t k, const jolt_fr_t *scalars, jolt_table **out);
/* Polynomial::evaluate (dense.rs:340-366): sum_x t[x] * eq(x, point) */
int32_t jolt_table_evaluate(jolt_ctx *ctx, const jolt_table *t, const jolt_fr_t *point, size_t n, jolt_fr_t *out);
/* Sum of all entries (input claims of linear members; DenseMember::with_sum, jolt-sumcheck/src/tests.rs:1129-1135) */
int32_t jolt_table_sum(jolt_ctx *ctx, const jolt_table *t, jolt_fr_t *out);

/* ------------------------------------------------------------------------------------------------------------
 * Sumcheck members: the device twin of NaiveSumcheckProver (crates/jolt-kernels/src/reference/naive.rs:53-377),
 * serving every cycle-domain relation of SURVEY.md section 8 a13 from one descriptor, plus the split-eq product
 * member (optimized/support.rs:391-411, ram_hamming_booleanity.rs:111-135).
 *
 * Summand = sum_k coeffs[k] * prod_{f in term k} table[factors[f]]   (jolt_claims::Expr, claims.rs:17-46 with
 * Challenge leaves pre-folded into coeffs).  A term with no factors is a constant.
 * ---------------------------------------------------------------------------------------------------------- */
#define JOLT_MAX_MEMBER_TABLES 40
#define JOLT_MAX_MEMBER_TERMS 16
#define JOLT_MAX_MEMBER_FACTORS 64
#define JOLT_MAX_DEGREE 7

typedef struct {
    uint32_t n_tables;
    uint32_t n_terms;
    uint32_t degree;              /* relation.degree(): round polynomials have degree+1 evaluations            */
    int32_t order;                /* JOLT_ORDER_*; every T-scale member of the reference binds LowToHigh        */
    const uint32_t *term_offsets; /* n_terms+1 entries into `factors`                                          */
    const uint32_t *factors;      /* table indices                                                             */
    const jolt_fr_t *coeffs;      /* n_terms                                                                   */
} jolt_member_desc;

/* NaiveSumcheckProver::new (naive.rs:136-205).  Takes OWNERSHIP of the tables (they are freed with the member). */
int32_t jolt_member_create_expr(jolt_ctx *ctx, jolt_table *const *tables, const jolt_member_desc *desc, jolt_member **out);

/* The optimized tier's fused summands as a descriptor rewrite instead of a new kernel:
 *   summand = sum_g prod_{f in group g} ( factor_consts[f] + sum_k lc_coeffs[k] * table[lc_tables[k]] )
 * e.g. inc_claim_reduction's A = s1*eq(p1) + s2*eq(p2) linear-leaf fusion
 * (crates/jolt-kernels/src/optimized/inc_claim_reduction.rs:68-89) is one factor with two entries.
 * flags bit 0 = evaluate t in {0,2,..,degree} only and let the caller recover s(1) = claim - s(0)
 * (optimized/support.rs:450-459 round_poly_from_skipped_evals); prove_round then returns `degree` values. */
#define JOLT_MEMBER_FLAG_SKIP_ONE 1u
/* flags bit 1 = BORROW: the member only READS the given tables (they stay owned by the caller and must outlive it) and
 * binds into scratch of its own (N/2 + N/4 entries per table).  One resident witness table can then serve every
 * relation that mentions it and the PCS opening afterwards, instead of one `oracle_table` materialisation per kernel
 * (crates/jolt-kernels/src/reference/views.rs:20-33).  Borrowing members can be rewound with jolt_member_reset. */
#define JOLT_MEMBER_FLAG_BORROW_TABLES 2u
typedef struct {
    uint32_t n_tables, n_groups, n_factors, n_lc;
    uint32_t degree;
    int32_t order;
    uint32_t flags;
    const uint32_t *group_factor_offsets; /* n_groups+1  */
    const uint32_t *factor_lc_offsets;    /* n_factors+1 */
    const jolt_fr_t *factor_consts;       /* n_factors, or NULL = all zero */
    const uint32_t *lc_tables;            /* n_lc */
    const jolt_fr_t *lc_coeffs;           /* n_lc */
} jolt_member_lc_desc;
int32_t jolt_member_create_lc(jolt_ctx *ctx, jolt_table *const *tables, const jolt_member_lc_desc *desc, jolt_member **out);
/* eq(w, j) * a(j) * b(j), eq served from split tables (GruenSplitEqPolynomial::new_with_scaling(w, LowToHigh, scale),
 * crates/jolt-poly/src/split_eq.rs:187-236).  Takes ownership of a and b. */
int32_t jolt_member_create_split_eq_product(jolt_ctx *ctx, jolt_table *a, jolt_table *b, const jolt_fr_t *w, size_t n,
                                            const jolt_fr_t *scale, jolt_member **out);
int32_t jolt_member_create_split_eq_product_borrowed(jolt_ctx *ctx, jolt_table *a, jolt_table *b, const jolt_fr_t *w, size_t n,
                                                     const jolt_fr_t *scale, jolt_member **out);
/* eq(w, j) * sum_{v<V} coeffs[v] * prod_{i<F} tables[v*F+i](j), eq served from split tables -- the optimized tier's form of
 * instruction_ra_virtualization / ram_ra_virtualization / ram_hamming_booleanity.  F in {2,3,4},
 * V <= 16, degree F+1.  prove_round returns the F eq-stripped sums q(0), q(2), .., q(F); aux_out = {current_scalar,
 * w[current_index-1], 0}; the caller recovers q(1) from the running claim and multiplies by the linear eq factor
 * (GruenSplitEqPolynomial::gruen_poly_from_evals, crates/jolt-poly/src/split_eq.rs:419-447).
 * flags: JOLT_MEMBER_FLAG_BORROW_TABLES.  shard_scale may be NULL (see the sharded product variant below). */
/* eq(w,j) * q(j) for an arbitrary inner summand q in jolt_member_lc_desc form (degree = the INNER degree dq): the optimized tier's
 * shape for every "eq times something" relation (GruenRoundMessage, crates/jolt-kernels/src/optimized/support.rs:340-412;
 * instruction_input.rs:1-22, instruction_claim_reduction.rs).  No T-sized eq table is read or bound; prove_round returns
 * q(0), q(2), .., q(dq) (dq values); the round polynomial is l(t)*q(t) (jolt_host_gruen_poly_from_q).  LowToHigh only.
 * shard_scale (may be NULL) as in jolt_member_create_split_eq_product_sharded.  final_values appends the bound eq scalar. */
int32_t jolt_member_create_split_eq_lc(jolt_ctx *ctx, jolt_table *const *tables, const jolt_member_lc_desc *desc, const jolt_fr_t *w, size_t n,
                                       const jolt_fr_t *scale, const jolt_fr_t *shard_scale, jolt_member **out);
/* The two constructors above over COMPACT-SCALAR polynomials: slot i is tables[i] (a field table) or ints[i] (a resident JOLT_INT_U64 witness column, borrowed:
 * it must outlive the member) -- exactly one of the two per slot.  Replaces the optimized tier's Polynomial<T> members: round 0 multiplies field elements with
 * machine integers through the deferred-reduction accumulator (FrSmallScalarAccumulator / mul_u64, crates/jolt-field/src/bn254/mont.rs:286-305,343-427) and the
 * FIRST bind produces the field tables (Polynomial::bind_to_field, crates/jolt-poly/src/dense.rs:129-142); no promotion pass, 8 instead of 32 bytes per entry in
 * round 0.  w == NULL: jolt_member_create_lc (flags of `desc` honoured, tables always borrowed); w != NULL: jolt_member_create_split_eq_lc.  LowToHigh only.
 * Identical round sums and final values to the member over promoted tables (jolt_table_from_ints).  Other integer kinds: JOLT_ERR_UNSUPPORTED. */
typedef struct jolt_ints jolt_ints;
int32_t jolt_member_create_lc_small(jolt_ctx *ctx, jolt_table *const *tables, const jolt_ints *const *ints, const jolt_member_lc_desc *desc, const jolt_fr_t *w,
                                    size_t n, const jolt_fr_t *scale, const jolt_fr_t *shard_scale, jolt_member **out);
/* test hook (CPU suite): the per-pair evaluation of the integer round kernel compiled for the host -- one LowToHigh pair of a member in `desc` form; slot i is an
 * integer column iff is_int[i] (entries int_pairs[2i], [2i+1]; otherwise fr_pairs[2i], [2i+1]); out[s], s < n_evals <= 4: the summand at 0, 1, 2, .. (skip_one: 0, 2, 3, ..) */
int32_t jolt_host_small_round_pair(const jolt_member_lc_desc *desc, const uint8_t *is_int, const uint64_t *int_pairs, const jolt_fr_t *fr_pairs, uint32_t n_evals,
                                   int32_t skip_one, jolt_fr_t *out);
int32_t jolt_member_create_split_eq_uniform(jolt_ctx *ctx, jolt_table *const *tables, uint32_t V, uint32_t F, const jolt_fr_t *coeffs,
                                            const jolt_fr_t *w, size_t n, const jolt_fr_t *scale, const jolt_fr_t *shard_scale,
                                            uint32_t flags, jolt_member **out);
/* Sharded variant (one process per GPU): this rank holds rows of the block selected by the top log2 G
 * variables; `w` are the n local coordinates (the low ones), `scale` the global initial scalar and `shard_scale` =
 * eq(w_hi, rank) multiplies the E_out tables so that the ranks' partial sums simply add. Borrows a and b. */
int32_t jolt_member_create_split_eq_product_sharded(jolt_ctx *ctx, jolt_table *a, jolt_table *b, const jolt_fr_t *w, size_t n,
                                                    const jolt_fr_t *scale, const jolt_fr_t *shard_scale, jolt_member **out);
/* Rewind a BORROW member to round 0 (no device work). */
int32_t jolt_member_reset(jolt_member *m);
/* New scaling factor for a split-eq member that has bound nothing yet (GruenSplitEqPolynomial::new_with_scaling,
 * crates/jolt-poly/src/split_eq.rs:180-215): lets a reset member serve the next proof with another eq scalar. */
int32_t jolt_member_set_scale(jolt_member *member, const jolt_fr_t *scale);
int32_t jolt_member_num_rounds(const jolt_member *m, size_t *rounds);
int32_t jolt_member_degree(const jolt_member *m, uint32_t *degree);

/* ProveRounds::prove_round (crates/jolt-sumcheck/src/prover.rs:57-66), device half: bind `bind` (NULL on the
 * member's first active round -- the fused contract, prover.rs:45-51) and return the round sums.
 *   expr member     : evals_out[t] = s(t), t = 0..degree           (n_evals must be degree+1)
 *   split-eq member : evals_out = { q(0), q(inf) }, the eq-stripped endpoints that
 *                     gruen_poly_deg_3 (split_eq.rs:383-417) completes on the host (n_evals must be 2);
 *                     aux_out (3 entries, may be NULL) receives {current_scalar, w[current_index-1], 0}.
 * The caller (Rust: UnivariatePoly::from_evals, univariate.rs:198-202) interpolates and checks s(0)+s(1). */
int32_t jolt_member_prove_round(jolt_member *m, const jolt_fr_t *bind, jolt_fr_t *evals_out, size_t n_evals,
                                jolt_fr_t *aux_out);
/* All members of one batch round with as few launches/syncs as possible -- the BuildRoundScheduler hook
 * (crates/jolt-kernels/src/backend.rs:68-70, RoundScheduler::batch_prove_round prover.rs:110-115).
 * binds[i] may be NULL; evals_out is the concatenation of each member's evals (same layout as above). */
int32_t jolt_round_group_prove(jolt_ctx *ctx, jolt_member *const *members, size_t n_members, const jolt_fr_t *const *binds,
                               jolt_fr_t *evals_out, size_t evals_capacity);
/* ProveRounds::finish_rounds (prover.rs:68-71) */
int32_t jolt_member_finish(jolt_member *m, const jolt_fr_t *bind);
/* RoundScheduler::batch_finish_rounds (prover.rs:116-119): the final binds of all members in as few launches as possible */
int32_t jolt_round_group_finish(jolt_ctx *ctx, jolt_member *const *members, size_t n_members, const jolt_fr_t *const *binds);
/* SumcheckKernel::output_claims (naive.rs:331-347): every table's fully bound value, in table order; the split-eq
 * member appends its bound eq scalar (k = n_tables (+1)).  JOLT_ERR_NOT_FULLY_BOUND before the last bind. */
int32_t jolt_member_final_values(jolt_member *m, jolt_fr_t *out, size_t k);
/* sum over the hypercube of the summand (the claim a stage would consume; test/bench convenience) */
int32_t jolt_member_input_claim(jolt_member *m, jolt_fr_t *out);
int32_t jolt_member_destroy(jolt_member *m);

/* ------------------------------------------------------------------------------------------------------------
 * G1 MSM and HyperKZG prover pieces.
 * ---------------------------------------------------------------------------------------------------------- */
/* Upload bases once and keep them affine on the device -- replaces the per-call `into_affine` of every base in
 * Bn254G1::msm (crates/jolt-crypto/src/ec/bn254/mod.rs:205).  An MSM over bases[..n] uses the prefix. */
int32_t jolt_srs_upload_g1(jolt_ctx *ctx, const jolt_g1_t *bases, size_t n, jolt_srs **out);
/* HyperKZGScheme::setup_from_secret (crates/jolt-hyperkzg/src/scheme.rs:54-73): g1_powers[i] = beta^i * g1,
 * generated on the device (count = max_degree + 1). */
int32_t jolt_srs_setup_from_secret(jolt_ctx *ctx, const jolt_fr_t *beta, size_t count, const jolt_g1_t *g1, jolt_srs **out);
int32_t jolt_srs_len(const jolt_srs *srs, size_t *n);
int32_t jolt_srs_download(jolt_ctx *ctx, const jolt_srs *srs, size_t offset, size_t n, jolt_g1_t *out); /* z = 1 */
int32_t jolt_srs_free(jolt_ctx *ctx, jolt_srs *srs);
/* Fixed-base tables for an SRS that outlives many MSMs (HyperKZGProverSetup.g1_powers is built once per setup,
 * crates/jolt-hyperkzg/src/types.rs:105-108, and every kzg_commit / kzg_open_batch MSM multiplies a prefix of it): keep
 * 2^(c*w) * srs[i] for every c-bit window w resident, so that all windows of an MSM share ONE bucket set and c can grow to 23-26 bits
 * (11 instead of 16 windows of point additions for 254-bit scalars: scalars above r/2 are negated, the top window is unsigned; DESIGN.md
 * section 3.5c).  Costs ceil(253/c) copies of the bases in HBM, stored in the limb-form Montgomery radix of the bucket kernels.
 * window_bits = 0 picks c from the SRS length (23 from 2^24 points, else <= 24); MSMs over fewer than min_terms terms (0 = 2^(c-2)) keep
 * the per-window method.  Results are the same points. */
int32_t jolt_srs_precompute_windows(jolt_ctx *ctx, jolt_srs *srs, uint32_t window_bits, size_t min_terms);

/* Measurement hook (no reference counterpart): HIP events around the dominant kernel of a fixed-base MSM (the bucket sums, k_fx_buckets_ordered) on the stream it is
 * launched on, and the number of mixed additions of that launch (the non-zero signed digits of its scalars).  bench.py's `roofline_msm` divides the two. */
int32_t jolt_msm_profile_buckets(jolt_ctx *ctx, int32_t enable);
int32_t jolt_msm_profile_buckets_last(jolt_ctx *ctx, float *ms, uint64_t *additions);
/* Measurement hook (no reference counterpart; SURVEY.md section 8d: "the run must print the peak it divides by"): the chip-wide issue rate of v_mad_u64_u32 -- the
 * instruction the bucket sums are bound by -- measured now, on this context's device: a register-only loop of independent 64-bit multiply-adds launched until
 * >= target_ms of kernel time have been timed with HIP events on the context's stream; *mads_per_s = the best launch's lane-operations per second.
 * timed_ms / launches (optional): what was timed. */
int32_t jolt_ctx_measure_mad_peak(jolt_ctx *ctx, float target_ms, double *mads_per_s, float *timed_ms, uint32_t *launches);

/* JoltGroup::msm(bases, scalars) (crates/jolt-crypto/src/ec/group.rs:63-70, bn254/mod.rs:195-212) with the bases
 * = srs[..n].  Length mismatch (n > srs length) is JOLT_ERR_SRS_TOO_SMALL (the Rust shim asserts equal lengths
 * before calling, keeping the reference's panic).  Scalars from host memory or from a device table. */
int32_t jolt_msm_g1(jolt_ctx *ctx, const jolt_srs *srs, const jolt_fr_t *scalars, size_t n, jolt_g1_t *out);
int32_t jolt_msm_g1_table(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *scalars, size_t n, jolt_g1_t *out);
/* the same for scalars the caller knows to be full-width field elements (a polynomial folded by a challenge): mid-length MSMs may then use the mid window-table set */
int32_t jolt_msm_g1_table_full_width(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *scalars, size_t n, jolt_g1_t *out);
/* Up to three prefix MSMs in flight together on the side lanes, collected later: the dense columns of CommitWitness::commit_witness
 * (crates/jolt-kernels/src/commitment.rs:137-160, one kzg_commit per committed polynomial) while the caller commits its one-hot columns
 * (jolt_grid_commit_onehot) on the main stream in between.  begin: JOLT_ERR_UNSUPPORTED, with nothing enqueued, for count > 3 or a context with fewer lanes (the caller
 * then commits one by one).  Between begin and finish no other MSM entry point of the context may be called (JOLT_ERR_INVALID_ARG); finish frees the handle
 * whatever it returns; out[i] = sum_{k < n[i]} scalars[i][k] * srs[k]. */
typedef struct jolt_msm_pending jolt_msm_pending;
int32_t jolt_msm_g1_tables_begin(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *const *scalars, const size_t *n, size_t count, jolt_msm_pending **out);
int32_t jolt_msm_g1_tables_finish(jolt_ctx *ctx, jolt_msm_pending *pending, jolt_g1_t *out);

/* fold_polynomials (crates/jolt-hyperkzg/src/scheme.rs:88-114): levels_out[0] = clone of evals, then ell-1
 * LowToHigh folds with point[ell-1], ..., point[1]; ell output tables of length 2^ell, ..., 2. */
int32_t jolt_hyperkzg_fold(jolt_ctx *ctx, const jolt_table *evals, const jolt_fr_t *point, size_t ell, jolt_table **levels_out);
/* v[t][j] = P_j(u_t) (eval_univariate, crates/jolt-hyperkzg/src/kzg.rs:51-59,84-85): v_out is 3*ell, row-major */
int32_t jolt_hyperkzg_eval3(jolt_ctx *ctx, jolt_table *const *levels, size_t ell, const jolt_fr_t u[3], jolt_fr_t *v_out);
/* B = sum_j q^j * P_j (kzg.rs:95-105), length = len(levels[0]) */
int32_t jolt_hyperkzg_rlc(jolt_ctx *ctx, jolt_table *const *levels, size_t ell, const jolt_fr_t *q, jolt_table **out);
/* compute_witness_polynomial (kzg.rs:34-46): h = f / (x - u), len(f)-1 entries (parallel scan of the Horner
 * recurrence, exact). */
int32_t jolt_hyperkzg_witness_poly(jolt_ctx *ctx, const jolt_table *f, const jolt_fr_t *u, jolt_table **out);

/* ------------------------------------------------------------------------------------------------------------
 * Host-side mirror (C++ in this repo because the image has no Rust toolchain): the reference's callers of the path
 * restated above the C ABI so that the parity tests read like the reference's own.  In a Rust deployment these stay
 * in jolt-sumcheck / jolt-hyperkzg; they are exported here for tests and bench only.
 * ---------------------------------------------------------------------------------------------------------- */
/* Host field helpers used by round-message assembly (no GPU needed). */
int32_t jolt_host_fr_mul(const jolt_fr_t *a, const jolt_fr_t *b, jolt_fr_t *out);
int32_t jolt_host_fr_add(const jolt_fr_t *a, const jolt_fr_t *b, jolt_fr_t *out);
int32_t jolt_host_fr_sub(const jolt_fr_t *a, const jolt_fr_t *b, jolt_fr_t *out);
int32_t jolt_host_fr_inv(const jolt_fr_t *a, jolt_fr_t *out);
int32_t jolt_host_fr_from_u64(uint64_t v, jolt_fr_t *out);
int32_t jolt_host_fr_mul_shifted(const jolt_fr_t *a, const jolt_fr_t *c, jolt_fr_t *out); /* c low limbs must be 0 */
/* EqPolynomial::evals_serial (crates/jolt-poly/src/eq.rs:299-315) on the host, n <= 20: the K-entry address tables of one-hot members */
int32_t jolt_host_eq_evals(const jolt_fr_t *r, size_t n, const jolt_fr_t *scale, jolt_fr_t *out);
/* the kernels' multiplication algorithm (nine 29-bit limbs, product scanning) compiled for the host; field 0 = Fr, 1 = Fq */
int32_t jolt_host_mul_limbs29(int32_t field, const jolt_fr_t *a, const jolt_fr_t *b, jolt_fr_t *out);
/* The limb-form Fq arithmetic of the MSM's bucket sums (csrc/fq_limb.hip.h: 2^261 Montgomery radix, lazy reduction, dedicated squaring,
 * two-product reduction) and its XYZZ mixed addition, compiled for the host for the CPU suite.  Operands and results are canonical Fq /
 * affine points in STANDARD Montgomery form.  op: 0 a*b, 1 a^2, 2 a*b + c*d, 3 (a - b)*c, 4 (a - b - 2c)*d. */
int32_t jolt_host_fq_limb_op(int32_t op, const jolt_fr_t *a, const jolt_fr_t *b, const jolt_fr_t *c, const jolt_fr_t *d, jolt_fr_t *out);
int32_t jolt_host_g1_sum_limb_form(const uint64_t *points, const uint8_t *negate, size_t count, jolt_g1_t *out);
/* The signed-digit recoding of the fixed-base MSM for one scalar (negation above r / 2, c-bit signed windows, unsigned top window), built for
 * the host: keys_out[w] = |digit_w| | sign << 31 for w < ceil(253 / c); *buckets_out = the bucket count of a table set with this c. */
int32_t jolt_host_fx_digits(const jolt_fr_t *scalar, uint32_t window_bits, uint32_t *keys_out, uint32_t *n_windows_out, uint32_t *buckets_out);
/* ... and the size of the REGION the capacity sort gives segment `segment` (256 buckets) of an n-term MSM over uniform scalars: expected digits + 8 standard deviations + 64
 * (msm_fixed.hip section 2d; the sort falls back to exact offsets on the device when a region overflows).  Host code, for the CPU suite. */
int32_t jolt_host_fx_segment_capacity(uint64_t n, uint32_t window_bits, uint32_t segment, uint32_t *capacity);
/* UnivariatePoly::from_evals / evaluate (crates/jolt-poly/src/univariate.rs:198-202) */
int32_t jolt_host_univariate_from_evals(const jolt_fr_t *evals, size_t n, jolt_fr_t *coeffs_out);
int32_t jolt_host_univariate_evaluate(const jolt_fr_t *coeffs, size_t n, const jolt_fr_t *x, jolt_fr_t *out);
/* s(t) = l(t) q(t) from q(0), q(2), .., q(dq) and the claim s(0)+s(1) (split_eq.rs:419-447 gruen_poly_from_evals, with the
 * evaluation set of the device kernel): dq + 2 coefficients. */
int32_t jolt_host_gruen_poly_from_q(const jolt_fr_t *current_scalar, const jolt_fr_t *point_i, const jolt_fr_t *q_evals, size_t dq,
                                    const jolt_fr_t *s0_plus_s1, jolt_fr_t *coeffs_out);
/* GruenSplitEqPolynomial::gruen_poly_deg_3 (split_eq.rs:383-417): 4 coefficients */
int32_t jolt_host_gruen_poly_deg_3(const jolt_fr_t *current_scalar, const jolt_fr_t *point_i, const jolt_fr_t *q_constant,
                                   const jolt_fr_t *q_quadratic, const jolt_fr_t *s0_plus_s1, jolt_fr_t *coeffs_out);
/* Booleanity address phase (stage 6a), host half: OptimizedBooleanityAddressKernel::{prove_round, bind} (crates/jolt-kernels/src/optimized/
 * booleanity.rs:320-398) over the K-entry pushforward masses of jolt_onehot_pushforward (K = 2^log_k_chunk = 16 .. 256 points: the reference
 * keeps this loop on the host, :46-49).  Tables are n_polys rows of `stride` entries with the first `len` live; weights[i] = gamma^(2i);
 * round returns s(0..3) for UnivariatePoly::from_evals; bind halves `len` (linear and eq as multilinears, squared with (1-r)^2 / r^2). */
int32_t jolt_host_booleanity_address_round(const jolt_fr_t *linear, const jolt_fr_t *squared, size_t n_polys, size_t stride, size_t len,
                                           const jolt_fr_t *weights, const jolt_fr_t *eq_address, jolt_fr_t *evals_out);
int32_t jolt_host_booleanity_address_bind(jolt_fr_t *linear, jolt_fr_t *squared, size_t n_polys, size_t stride, size_t len, jolt_fr_t *eq_address,
                                          const jolt_fr_t *challenge);
/* Hamming-weight claim reduction (stage 7), host half: HammingWeightKernel (crates/jolt-kernels/src/optimized/hamming_weight_claim_reduction.rs:150-300) over the
 * K_chunk-entry pushforward masses G_i of all RA columns (jolt_onehot_pushforward against eq(r_cycle, .): the relation's one T-scale pass, :83-117).
 *   weights: W_i(k) = gamma^(3i) + gamma^(3i+1) eq(r_address, k) + gamma^(3i+2) eq(virtualization_points[i], k)  (:187-207), n_polys rows of 2^log_k entries;
 *   round:   evals_out = {s(0), s(2), sum_i sum_k G_i(k) W_i(k)} of the summand sum_i G_i W_i (group_evals :255-266; the caller recovers s(1) from the running
 *            claim, round_poly_from_skipped_evals; the third value is the input claim before the first round);
 *   bind:    every table bound as a multilinear, low variable first (:243-252); the output claims are the bound G_i.
 * Tables are n_polys rows of `stride` entries with the first `len` live. */
int32_t jolt_host_hamming_weights(const jolt_fr_t *gamma, const jolt_fr_t *r_address, const jolt_fr_t *virtualization_points, size_t n_polys, size_t log_k,
                                  jolt_fr_t *out);
int32_t jolt_host_pair_tables_round(const jolt_fr_t *g, const jolt_fr_t *w, size_t n_polys, size_t stride, size_t len, jolt_fr_t *evals_out /* 3 */);
int32_t jolt_host_pair_tables_bind(jolt_fr_t *g, jolt_fr_t *w, size_t n_polys, size_t stride, size_t len, const jolt_fr_t *challenge);
/* The Fiat-Shamir surface of members the caller drives round by round outside prove_batch (RamReadWriteKernel's rounds, the read-RAF
 * phases): Transcript::{new, append_bytes, append, challenge, challenge_scalar, state} (crates/jolt-transcript/src/legacy.rs:32-100) and the
 * reference's encodings of what the path absorbs -- a field element as 32 BIG-endian bytes (legacy.rs:116-123), Label / LabelWithCount / U64Word
 * (legacy.rs:146-209), a compressed labelled round polynomial (crates/jolt-sumcheck/src/round_proof.rs:129-143: LabelWithCount(label, n - 1), the
 * constant, the coefficients of degree >= 2).  Three engines; every `transcript_label` argument of this header selects one with its two top bits:
 *   JOLT_TRANSCRIPT_TEST (0)            the deterministic test transcript of rounds 1-5;
 *   JOLT_TRANSCRIPT_BLAKE2B_LEGACY (1)  jolt_transcript::LegacyBlake2bTranscript = DigestTranscript<Blake2b<U32>> (crates/jolt-transcript/src/digest.rs:84-189),
 *                                       the transcript of the reference's benchmark profile (crates/jolt-prover/src/profile.rs:69,726);
 *   JOLT_TRANSCRIPT_KECCAK_SPONGE (2)   jolt_transcript::KeccakTranscript = SpongeTranscript<spongefish Keccak> (legacy.rs:211-305), pinned by the reference's
 *                                       known-answer vector (crates/jolt-transcript/tests/keccak_tests.rs:13-29).
 *   JOLT_TRANSCRIPT_BLAKE2B_SPONGE (3)  jolt_transcript::Blake2bTranscript = SpongeTranscript<spongefish Blake2b512> (lib.rs:63-66); the reference's vector
 *                                       (tests/blake2b_tests.rs:13-37) pins it through the first challenge only -- unpinned beyond, used by no parity claim.
 * For engines 1 - 3 an integer label L names the session "jolt-amd/<L mod 2^62>" (ASCII); _create_labelled takes the reference's byte labels (b"Jolt").
 * A Rust caller keeps its own transcript object and never calls these (jolt_round_transcript_fn / jolt_open_transcript_fn are its hooks). */
enum { JOLT_TRANSCRIPT_TEST = 0, JOLT_TRANSCRIPT_BLAKE2B_LEGACY = 1, JOLT_TRANSCRIPT_KECCAK_SPONGE = 2, JOLT_TRANSCRIPT_BLAKE2B_SPONGE = 3 };
#define JOLT_TRANSCRIPT_LABEL(kind, label) (((uint64_t)(kind) << 62) | ((uint64_t)(label) & (((uint64_t)1 << 62) - 1)))
typedef struct jolt_host_transcript jolt_host_transcript;
int32_t jolt_host_transcript_create(uint64_t label, jolt_host_transcript **out);
int32_t jolt_host_transcript_create_labelled(int32_t kind, const uint8_t *label, size_t label_len, jolt_host_transcript **out); /* label_len <= 32 (MAX_LABEL_LEN) */
int32_t jolt_host_transcript_append_fr(jolt_host_transcript *t, const jolt_fr_t *values, size_t count); /* 32 big-endian bytes each, one append per value */
int32_t jolt_host_transcript_append_bytes(jolt_host_transcript *t, const uint8_t *bytes, size_t count); /* e.g. a compressed G1 point (32 bytes) */
int32_t jolt_host_transcript_append_label(jolt_host_transcript *t, const char *label, int32_t with_count, uint64_t count); /* Label (<= 32 bytes) / LabelWithCount (<= 24) */
int32_t jolt_host_transcript_append_u64_word(jolt_host_transcript *t, uint64_t value);
int32_t jolt_host_transcript_append_round_poly(jolt_host_transcript *t, const char *label, const jolt_fr_t *coefficients, size_t count); /* all `count` coefficients in; the linear one is skipped */
int32_t jolt_host_transcript_challenge(jolt_host_transcript *t, int32_t full_width, jolt_fr_t *out);    /* 0: 125-bit challenge shape */
int32_t jolt_host_transcript_state(const jolt_host_transcript *t, uint8_t *out32);
int32_t jolt_host_transcript_destroy(jolt_host_transcript *t);
/* the two hash primitives behind engines 1 and 2, for known-answer tests: BLAKE2b (RFC 7693, unkeyed, 1..64 digest bytes), Keccak-f[1600] on a 200-byte state */
int32_t jolt_host_blake2b(const uint8_t *in, size_t n, size_t outlen, uint8_t *out);
int32_t jolt_host_keccak_f1600(uint8_t *state200);
/* G1 helpers on the host: group law, equality as points, compressed serialisation
 * (crates/jolt-crypto/src/ec/bn254/mod.rs:139-171) */
int32_t jolt_host_g1_add(const jolt_g1_t *p, const jolt_g1_t *q, jolt_g1_t *out);
int32_t jolt_host_g1_eq(const jolt_g1_t *p, const jolt_g1_t *q, int32_t *equal);
/* canonical coordinates on y^2 = x^3 + 3 (Jacobian: Y^2 = X^3 + 3 Z^6) or the identity: the check applied to points supplied by the caller (known level commitments) */
int32_t jolt_host_g1_is_on_curve(const jolt_g1_t *p, int32_t *on_curve);
int32_t jolt_host_g1_serialize_compressed(const jolt_g1_t *p, uint8_t out[32]);

/* jolt_sumcheck::prove_batch (crates/jolt-sumcheck/src/prover.rs:193-362) over device members with the
 * deterministic test transcript of oracle/mock_transcript.h (spec in that header; re-implemented, not shared).
 *   out_polys         max_num_vars x (max_degree+1) batched coefficients
 *   out_challenges    max_num_vars
 *   out_member_claims n_members,  out_final_claim 1
 * challenge_mode 0 = Transcript::challenge (125-bit), 1 = challenge_scalar (full width). */
int32_t jolt_host_prove_batch(jolt_ctx *ctx, jolt_member *const *members, size_t n_members, const jolt_fr_t *input_claims,
                              const jolt_fr_t *coefficients, const size_t *offsets, size_t max_num_vars, size_t max_degree,
                              uint64_t transcript_label, int32_t challenge_mode, int32_t use_round_group,
                              jolt_fr_t *out_polys, jolt_fr_t *out_challenges, jolt_fr_t *out_member_claims,
                              jolt_fr_t *out_final_claim);
/* Resumable form of the same round loop for the hypercube-sharded prover: local round sums come
 * from the device members (local_fn == NULL) or from a callback, are all-gathered over the ranks through `gather` and added
 * mod r; the loop can be paused after the shard-local rounds and resumed on members built from the gathered tables.
 * kinds: 0 = expr (degree+1 sums), 1 = expr with skipped s(1) (degree sums), 2 = split-eq product (2 sums: q(0), q(inf) -> gruen_poly_deg_3),
 * 3 = split-eq uniform product or eq-weighted member (degrees[i] - 1 sums q(0), q(2), .., q(degree-1) -> gruen_poly_from_q). */
typedef struct jolt_batch jolt_batch;
typedef int32_t (*jolt_local_round_fn)(void *user, const size_t *active, size_t n_active, const jolt_fr_t *const *binds,
                                       jolt_fr_t *evals_out, size_t evals_count);
typedef int32_t (*jolt_gather_fn)(void *user, const jolt_fr_t *local, size_t count, jolt_fr_t *gathered /* world*count */);
int32_t jolt_host_batch_begin(jolt_ctx *ctx, size_t n_members, const jolt_fr_t *input_claims, const jolt_fr_t *coefficients,
                              const size_t *rounds, const size_t *offsets, const int32_t *kinds, const uint32_t *degrees,
                              const jolt_fr_t *const *split_eq_points, const jolt_fr_t *split_eq_scales, size_t max_num_vars,
                              size_t max_degree, uint64_t transcript_label, int32_t challenge_mode, jolt_batch **out);
int32_t jolt_host_batch_run(jolt_batch *b, jolt_member *const *members, size_t n_rounds, int32_t world, jolt_gather_fn gather,
                            jolt_local_round_fn local_fn, void *user);
/* The batch's Fiat-Shamir: by default the deterministic test transcript; with a callback the CALLER's transcript
 * (ClearSumcheckRecorder::absorb_round, crates/jolt-sumcheck/src/recorder.rs:118-130): every round the loop hands over the batched
 * round polynomial in compressed form (coefficients without the linear term) and takes the challenge back.  All ranks of a sharded
 * batch must return the same challenge (they absorb the same summed polynomial).  fn = NULL restores the test transcript. */
typedef int32_t (*jolt_round_transcript_fn)(void *user, const jolt_fr_t *compressed_coeffs, size_t n_coeffs, jolt_fr_t *challenge_out);
int32_t jolt_host_batch_set_transcript(jolt_batch *b, jolt_round_transcript_fn fn, void *user);
int32_t jolt_host_batch_flush_binds(jolt_batch *b, jolt_member *const *members, jolt_fr_t *binds_out, int32_t *has_bind_out);
int32_t jolt_host_batch_split_eq_scalar(const jolt_batch *b, size_t member, jolt_fr_t *out);
int32_t jolt_host_batch_end(jolt_batch *b, jolt_fr_t *out_polys, jolt_fr_t *out_challenges, jolt_fr_t *out_member_claims,
                            jolt_fr_t *out_final_claim);
/* SplitLt (crates/jolt-kernels/src/optimized/support.rs:640-760): LT(., r_cycle) + constant from ~sqrt(T) split tables, bound
 * low-to-high; values equal the dense jolt_lt_evals table (plus the constant) bound identically.  constant may be NULL. */
typedef struct jolt_split_lt jolt_split_lt;
int32_t jolt_split_lt_create(jolt_ctx *ctx, const jolt_fr_t *r_cycle, size_t n, const jolt_fr_t *constant, jolt_split_lt **out);
int32_t jolt_split_lt_bind(jolt_ctx *ctx, jolt_split_lt *s, const jolt_fr_t *r);
int32_t jolt_split_lt_len(const jolt_split_lt *s, size_t *len);
int32_t jolt_split_lt_to_dense(jolt_ctx *ctx, const jolt_split_lt *s, jolt_table **out); /* all current evaluations (pair(y) for every y) */
int32_t jolt_split_lt_final_value(jolt_ctx *ctx, const jolt_split_lt *s, jolt_fr_t *out); /* JOLT_ERR_NOT_FULLY_BOUND before the last bind */
int32_t jolt_split_lt_free(jolt_ctx *ctx, jolt_split_lt *s);

/* sum_k a[k]*b[k] with ONE deferred Montgomery reduction per block of products: the deferred-reduction accumulator the round
 * kernels use for sums of products (WideAccumulator, crates/jolt-field/src/bn254/mont.rs:334-602; Accumulator contract
 * crates/jolt-field/src/algebra.rs:362-433).  Host build of the kernels' code; the value equals the plain field sum. */
int32_t jolt_host_fr_wide_dot(const jolt_fr_t *a, const jolt_fr_t *b, size_t n, jolt_fr_t *out);
/* The same accumulator ON THE DEVICE: sum_i a[i] * b[i] over two tables with plain field sums (deferred = 0) or per-thread
 * unreduced 512-bit sums of products reduced once per block of products (deferred = 1) -- equal canonical values
 * (Accumulator contract, crates/jolt-field/src/algebra.rs:362-433). */
int32_t jolt_table_dot(jolt_ctx *ctx, const jolt_table *a, const jolt_table *b, int32_t deferred, jolt_fr_t *out);

/* One-hot (Twist/Shout) selector columns as per-cycle hot indices.  Replaces ChunkIndexSource /
 * LazyFoldedRa (crates/jolt-kernels/src/optimized/lazy_ra.rs:39-268) and the pushforward G tables of the booleanity address
 * phase (optimized/booleanity.rs:24-31).  indices[p*cycles + j] in [0, k) or 0xFF on a cold cycle; k <= 255 (jolt_onehot_upload16 beyond). */
typedef struct jolt_onehot jolt_onehot;
int32_t jolt_onehot_upload(jolt_ctx *ctx, const uint8_t *indices, size_t n_polys, size_t cycles, uint32_t k, jolt_onehot **out);
/* The same with 16-bit indices (0xFFFF on a cold cycle, k <= 1024): the K = 256 chunks of long traces (log_k_chunk = 8 at log T >= 25,
 * crates/jolt-prover/src/config.rs:175-186), where index 255 is a valid address.  Every operator below accepts either width; the
 * LDS-staged and pair-table kernels of the lazily bound members are K <= 16 specialisations and are simply not selected. */
int32_t jolt_onehot_upload16(jolt_ctx *ctx, const uint16_t *indices, size_t n_polys, size_t cycles, uint32_t k, jolt_onehot **out);
int32_t jolt_onehot_download16(jolt_ctx *ctx, const jolt_onehot *source, uint16_t *out /* n_polys * cycles */);
int32_t jolt_onehot_free(jolt_ctx *ctx, jolt_onehot *source);
/* dense address-folded column: out[j] = scale_table[index(poly, j)], zero on cold cycles (the N x T "direct shape") */
int32_t jolt_onehot_materialize(jolt_ctx *ctx, const jolt_onehot *source, size_t poly, const jolt_table *scale_table, jolt_table **out);
/* out[p*k + a] = sum_j weights[j] * [index(p, j) == a]   (G_p = pushforward of the cycle weights to the address domain) */
int32_t jolt_onehot_pushforward(jolt_ctx *ctx, const jolt_onehot *source, const jolt_table *weights, jolt_table **out);
/* Member eq(w,j) * sum_v coeffs[v] * prod_{i<F} ra_{vF+i}(j), ra_p(j) = scale_tables[p*k + index(p,j)] (host array n_polys*k), with
 * the selector columns bound lazily: rounds 0..3 gather through the indices, the fourth bind materialises cycles/16-entry tables
 * (LazyFoldedRa::bind).  Round sums, binds and final values are those of jolt_member_create_split_eq_uniform over the
 * materialised columns, bit for bit.  n = log2(cycles) >= 4.  The source must outlive the member. */
int32_t jolt_member_create_lazy_ra_uniform(jolt_ctx *ctx, const jolt_onehot *source, const jolt_fr_t *scale_tables, uint32_t V, uint32_t F,
                                           const jolt_fr_t *coeffs, const jolt_fr_t *w, size_t n, const jolt_fr_t *scale, jolt_member **out);

/* One shard of a hypercube-sharded batch: the source holds this rank's block of cycles, w its n local coordinates, shard_scale =
 * eq(w_hi, rank) (as jolt_member_create_split_eq_product_sharded).  After the fourth bind the tables (what
 * jolt_round_group_pack_tables hands over) carry coeffs[v] folded into the first factor of product v. */
int32_t jolt_member_create_lazy_ra_uniform_sharded(jolt_ctx *ctx, const jolt_onehot *source, const jolt_fr_t *scale_tables, uint32_t V, uint32_t F,
                                                   const jolt_fr_t *coeffs, const jolt_fr_t *w, size_t n, const jolt_fr_t *scale,
                                                   const jolt_fr_t *shard_scale, jolt_member **out);
/* Packed typed witness rows: ONE upload of the per-cycle records the committed columns derive from
 * (CommittedColumnsWitness, crates/jolt-kernels/src/commitment.rs:25-32; InstructionCycleRow) instead of a materialised field
 * column per polynomial; columns are expanded on the device.  Fields are little-endian integers inside a row of row_bytes bytes. */
typedef struct jolt_rows jolt_rows;
int32_t jolt_rows_upload(jolt_ctx *ctx, const void *rows, size_t n_rows, size_t row_bytes, jolt_rows **out);
/* The same copy in flight beside the context's kernels: the NEXT proof's rows moving over the link while the current proof runs (the witness is produced by the tracer
 * ahead of the prover, crates/jolt-witness/src/consumer.rs:129-143).  `rows` must be page-locked (jolt_host_pinned_alloc) and unchanged until _wait returned; _begin
 * returns at once, _wait blocks the host until the copy has landed (begun a proof earlier, it has); the handle then behaves like jolt_rows_upload's. */
int32_t jolt_rows_upload_begin(jolt_ctx *ctx, const void *rows, size_t n_rows, size_t row_bytes, jolt_rows **out);
int32_t jolt_rows_upload_wait(jolt_ctx *ctx, jolt_rows *rows);
/* Page-locked host memory for the row buffer the tracer fills (the Vec<CycleRow> of crates/jolt-host/src/program.rs's trace output, packed): jolt_rows_upload
 * from such a block runs at the link rate (no staging copy by the runtime).  Ordinary host memory in every other respect. */
int32_t jolt_host_pinned_alloc(jolt_ctx *ctx, size_t bytes, void **out);
int32_t jolt_host_pinned_free(jolt_ctx *ctx, void *p);
int32_t jolt_rows_free(jolt_ctx *ctx, jolt_rows *rows);
/* integer field (width 1/2/4/8 bytes at `offset`, optionally two's complement) -> Fr table (Polynomial::bind_to_field promotion) */
int32_t jolt_table_from_rows(jolt_ctx *ctx, const jolt_rows *rows, size_t offset, uint32_t width, int32_t is_signed, jolt_table **out);
/* the same field as a resident integer column (JOLT_INT_U64, or JOLT_INT_I64 sign-extended): the compact scalars of Polynomial<T> (crates/jolt-poly/src/dense.rs:
 * 129-142) straight from the uploaded rows -- what jolt_member_create_lc_small and the *_small operators read; freed with jolt_ints_free */
int32_t jolt_ints_from_rows(jolt_ctx *ctx, const jolt_rows *rows, size_t offset, uint32_t width, int32_t is_signed, jolt_ints **out);
/* ... and n_fields fields in ONE pass over the rows (a workgroup stages whole rows in LDS, every field is written from there): out[k] = field k.  The witness hand-over
 * of a proof extracts every typed column of WitnessBundle::from_row (crates/jolt-kernels/src/optimized/rows.rs:22-72) -- field by field the rows were read once per field. */
int32_t jolt_ints_from_rows_many(jolt_ctx *ctx, const jolt_rows *rows, const size_t *offsets, const uint32_t *widths, const int32_t *is_signed, size_t n_fields,
                                 jolt_ints **out);
/* n_polys hot-index columns from ONE address field (<= 16 bytes): index_i = (field >> shifts[i]) & (2^log_k - 1)
 * (RaChunkSelector::chunk_u128, crates/jolt-witness/src/witnesses/one_hot.rs:14-52); a row whose byte at valid_offset is 0 is a
 * cold cycle (Option::None, e.g. no RAM access); valid_offset = SIZE_MAX: every row is hot.  log_k <= 8 (log_k = 8 gives a 16-bit source). */
int32_t jolt_onehot_from_rows(jolt_ctx *ctx, const jolt_rows *rows, size_t offset, uint32_t width, const uint32_t *shifts, size_t n_polys,
                              uint32_t log_k, size_t valid_offset, jolt_onehot **out);
int32_t jolt_onehot_download(jolt_ctx *ctx, const jolt_onehot *source, uint8_t *out /* n_polys * cycles */);
/* The extractors' one-row lookahead window over a physical trace shorter than the padded cycle domain (RandomAccessRows::window /
 * WitnessBundle::from_row(current, next, env), crates/jolt-kernels/src/optimized/rows.rs:22-72): out[j], j < cycles, reads the field of
 * row j (lookahead = 0) or row j + 1 (lookahead = 1); rows at or beyond n_rows are padding rows (the field of TraceRow::default():
 * padding_value), and the last cycle has no next row (none_value).  n_rows > cycles is JOLT_ERR_SIZE_MISMATCH (rows.rs:44-53). */
int32_t jolt_table_from_rows_window(jolt_ctx *ctx, const jolt_rows *rows, size_t offset, uint32_t width, int32_t is_signed, int32_t lookahead,
                                    size_t cycles, int64_t padding_value, int64_t none_value, jolt_table **out);
/* Hot-index columns from a sentinel-packed address field (InstructionCycleRow::{pc_plus_one, ram_address_plus_one},
 * crates/jolt-kernels/src/optimized/instruction_read_raf.rs:82-123): 0 = no access (cold cycle), v > 0 = address v - 1 whose chunks
 * index_i = ((v - 1) >> shifts[i]) & (2^log_k - 1) become the columns; cycles beyond the physical rows are cold. */
int32_t jolt_onehot_from_rows_sentinel(jolt_ctx *ctx, const jolt_rows *rows, size_t offset, uint32_t width, const uint32_t *shifts, size_t n_polys,
                                       uint32_t log_k, size_t cycles, jolt_onehot **out);

/* Booleanity cycle phase over the same lazily bound columns (crates/jolt-kernels/src/optimized/booleanity.rs:436-633):
 * eq(w,j) * sum_i (H_i(j)^2 - rho[i]*H_i(j)), H_i(j) = scale_tables[i*k + index(i,j)] (the caller passes the gamma^i-pre-scaled
 * address tables and rho[i] = gamma^i).  prove_round returns (q(0), q(inf)) of the inner quadratic; the cubic is
 * gruen_poly_deg_3 on the host.  final_values: the bound H_i (the host divides by rho[i], booleanity.rs:652-657) + the eq scalar. */
int32_t jolt_member_create_lazy_booleanity(jolt_ctx *ctx, const jolt_onehot *source, const jolt_fr_t *scale_tables, const jolt_fr_t *rho,
                                           const jolt_fr_t *w, size_t n, const jolt_fr_t *scale, jolt_member **out);

/* Dory tier-1 (G1) streaming row commitments -- SURVEY.md section 8(f) row 2.  Replaces the MSM work of
 * DoryScheme::{feed_u64, feed_i128, feed_i128_rows_with} (crates/jolt-dory/src/streaming.rs:115-205: one ark msm_u64 / msm_i128
 * per row_width window over the first row_width G1 bases) and of process_one_hot_chunk(s_with) -> one_hot_chunk_commitments
 * (:230-275, :366-419: per chunk, commitment[k] = sum of the bases of the columns whose hot row is k).  The tier-2 pairing
 * product stays with the caller.  Integers are little-endian machine integers (i128 = two u64, low first, two's complement). */
enum { JOLT_INT_U64 = 0, JOLT_INT_I64 = 1, JOLT_INT_I128 = 2 };
int32_t jolt_ints_upload(jolt_ctx *ctx, const void *host, int32_t kind, size_t count, jolt_ints **out);
int32_t jolt_ints_free(jolt_ctx *ctx, jolt_ints *values);
/* out[r] = sum_j values[r*row_width + j] * srs[j] for the count / row_width rows, in row order.  row_width must be a power of
 * two (JOLT_ERR_INVALID_ARG, streaming.rs:99-102), <= the SRS length (JOLT_ERR_SRS_TOO_SMALL, :103-108) and divide count
 * (JOLT_ERR_SIZE_MISMATCH, :192-195). */
int32_t jolt_dory_commit_rows(jolt_ctx *ctx, const jolt_srs *srs, const jolt_ints *values, size_t row_width, jolt_g1_t *out);
/* out[chunk*k + row] for the cycles / chunk_width chunks of hot-index column `poly`; an empty row gives the identity
 * (Bn254G1::default(), streaming.rs:409-418).  Same argument checks as above (:376-392). */
int32_t jolt_dory_commit_onehot(jolt_ctx *ctx, const jolt_srs *srs, const jolt_onehot *source, size_t poly, size_t chunk_width, jolt_g1_t *out);

/* Promotion of device-resident integers (entries [offset, offset+len) of `values`) to a field table: Ring::from_u64 / from_i64 /
 * from_i128 per entry (crates/jolt-field/src/bn254/mod.rs:265-328), the From<T> of Polynomial<T>::bind_to_field (dense.rs:129-142)
 * for witness columns that already sit in HBM (no host round trip; jolt_table_from_u64 is the same from host memory). */
int32_t jolt_table_from_ints(jolt_ctx *ctx, const jolt_ints *values, size_t offset, size_t len, jolt_table **out);

/* The committed trace polynomials over the proof's shared commitment grid, for a KZG-type scheme (HyperKZG).  Every witness
 * polynomial is committed in one grid of log_k + log_t variables (crates/jolt-kernels/src/commitment.rs:1-8,86-130); cycle-major
 * placement: coefficient (address k, cycle j) at index k*T + j, dense columns at k = 0 (TracePlacement / TraceOpeningPoly::entry,
 * crates/jolt-kernels/src/optimized/opening.rs:340-372,404-420).
 *   jolt_grid_commit_onehot: out[p] = sum over the hot cycles j of column p of srs[hot_p(j)*T + j] -- kzg_commit
 *     (crates/jolt-hyperkzg/src/kzg.rs:15-27) of a polynomial whose coefficients are 0/1 with one 1 per hot cycle: additions only.
 *     source->k * T > srs length is JOLT_ERR_SRS_TOO_SMALL.
 *   jolt_grid_joint_polynomial: the joint polynomial of HomomorphicBatch::prove_batch (crates/jolt-openings/src/schemes.rs:487-524,
 *     RlcSource::to_dense crates/jolt-poly/src/multilinear.rs:159-170) over the 2^log_k * T grid:
 *     out[k*T + j] = sum_p onehot_scalars[p] * [hot_p(j) == k] + [k == 0] * sum_d dense_scalars[d] * dense[d][j];
 *     onehot_scalars runs over the polynomials of sources[0], sources[1], ... in order.  <= 4 sources, <= 8 dense columns. */
int32_t jolt_grid_commit_onehot(jolt_ctx *ctx, const jolt_srs *srs, const jolt_onehot *source, jolt_g1_t *out /* n_polys */);
int32_t jolt_grid_joint_polynomial(jolt_ctx *ctx, const jolt_onehot *const *sources, size_t n_sources, const jolt_fr_t *onehot_scalars,
                                   jolt_table *const *dense, size_t n_dense, const jolt_fr_t *dense_scalars, uint32_t log_k,
                                   jolt_table **out);

/* Instruction read+RAF checking (stage 5) -- SURVEY.md section 8(f) row 4.  Replaces the T-scale scans of
 * OptimizedInstructionReadRafKernel (crates/jolt-kernels/src/optimized/instruction_read_raf.rs): per prefix-suffix phase the
 * condensation of the per-cycle mass (:750-758), the fused RAF scan (:770-812) and the per-table suffix accumulators
 * (init_suffix_tables :901-971), and after the address rounds the combined-value / ra_i columns of the cycle rounds
 * (pending_combined_base / pending_ra_base :1203-1232; sum them with jolt_member_create_split_eq_lc: one group of 1 + ra_count factors).
 * The 256-entry prefix polynomials, checkpoints and address-round messages are host code (jolt_host_read_raf_address_* below); the flag claims of output_claims are
 * jolt_onehot_pushforward over the table-index column.
 *   rows: lookup_index as (lo, hi) u64 pairs, table_index (0xFF = no lookup table; < n_tables <= 126), raf_flag (0 / 1).
 *   suffix_offsets[n_tables + 1] / suffix_kinds[]: LookupTableKind::suffixes() of every table, flattened; a kind is the discriminant of
 *     `enum Suffixes` (crates/jolt-lookup-tables/src/tables/suffixes/mod.rs:120-170, all 48 built; XLEN = 64).
 *   jolt_read_raf_phase_scan: raf_out[q * 256 + chunk], q = left, right, identity, shift_half, shift_full, upper_all_ones (raw sums: the
 *     caller applies mul_pow_2 to the shift sums, :814-823; upper_all_ones only with canonical != 0, the `akita` feature);
 *     suffix_out[(suffix_offsets[t] + s) * 256 + chunk]; chunk = (lookup_index >> suffix_len) & 255, suffix_len = address_bits - 8 (phase + 1).
 *   jolt_read_raf_condense: u[j] *= v_table[(lookup_index[j] >> shift) & 255].
 *   jolt_read_raf_cycle_tables: v_tables = the `phases` bound-challenge eq tables (256 entries each), phases * 8 = address_bits. */
typedef struct jolt_read_raf jolt_read_raf;
int32_t jolt_read_raf_create(jolt_ctx *ctx, const uint64_t *lookup_index, const uint8_t *table_index, const uint8_t *raf_flag, size_t cycles, uint32_t n_tables,
                             jolt_read_raf **out);
int32_t jolt_read_raf_destroy(jolt_ctx *ctx, jolt_read_raf *rr);
int32_t jolt_read_raf_cycles(const jolt_read_raf *rr, size_t *cycles, uint32_t *n_tables);
int32_t jolt_read_raf_phase_scan(jolt_ctx *ctx, jolt_read_raf *rr, const jolt_table *u, uint32_t suffix_len, uint32_t address_bits, int32_t canonical,
                                 const uint32_t *suffix_offsets, const uint8_t *suffix_kinds, jolt_fr_t *raf_out, jolt_fr_t *suffix_out);
int32_t jolt_read_raf_condense(jolt_ctx *ctx, jolt_read_raf *rr, jolt_table *u, const jolt_fr_t *v_table, uint32_t shift);
int32_t jolt_read_raf_cycle_tables(jolt_ctx *ctx, jolt_read_raf *rr, const jolt_fr_t *table_values, const jolt_fr_t *raf_interleaved, const jolt_fr_t *raf_identity,
                                   const jolt_fr_t *v_tables, uint32_t phases, uint32_t address_bits, uint32_t ra_count, jolt_table **combined_out,
                                   jolt_table **ra_out);
/* suffix_mle.hip.h (Suffixes::suffix_mle for the kind's discriminant) built for the host, for the CPU suite; bits are masked to `len`. */
int32_t jolt_host_suffix_mle(uint32_t kind, uint64_t lo, uint64_t hi, uint32_t len, uint64_t *out);

/* The address rounds of the same kernel (host code, no device work): the 256-entry prefix polynomials of a phase, the per-table `combine`, the four RAF
 * decompositions, the round messages and binds, checkpoints.  Replaces instruction_read_raf.rs address_message (:973-1050), the address branch of bind
 * (:1235-1282), the prefix half of init_phase (:824-898) and init_cycle_rounds' table values (:1140-1160), and what they call in jolt-lookup-tables:
 * Prefixes::{default_checkpoint, evaluate} (tables/prefixes/mod.rs:236-250 + the 44 prefix files) and LookupTableKind::{prefixes, suffixes, combine}
 * (tables/mod.rs:228-246 + the 42 table files).  Table / prefix / suffix ids are the discriminants of `enum LookupTableKind` (tables/mod.rs:121-166),
 * `enum Prefixes` (prefixes/mod.rs:104-154) and `enum Suffixes`; XLEN = 64, phases of 8 address bits.
 *   jolt_lookup_suffix_layout: offsets[43] / kinds[]: the suffix_offsets / suffix_kinds arguments of jolt_read_raf_phase_scan for n_tables = 42 real tables.
 *   per phase p = 0 .. 15:  jolt_read_raf_phase_scan (device) -> jolt_host_read_raf_address_init_phase(p, raf_out, suffix_out) -> 8 x { message, bind };
 *     message: evals_out = s(0), s(1) = previous_claim - s(0), s(2) (UnivariatePoly::from_evals order); bind: *phase_done = 1 after the 8th, when
 *     jolt_host_read_raf_address_v_table(p) is eq(phase challenges, .) for jolt_read_raf_condense / jolt_read_raf_cycle_tables.
 *     message with previous_claim == NULL sums s(1) from the tables: in round 0, s(0) + s(1) is then the relation's input claim.
 *   jolt_host_read_raf_address_prove_phase: the 8 rounds of the open phase in one call (message -> from_evals coefficients c0, c1, c2 -> transcript ->
 *     bind); the transcript is the caller's jolt_round_transcript_fn or, with fn == NULL, the library's TEST transcript (append c0..c2, Transcript::challenge).
 *   jolt_host_read_raf_address_finish: after phase 15: table_values[42], raf_interleaved, raf_identity = the arguments of jolt_read_raf_cycle_tables.
 *   canonical != 0 is the `akita` feature's upper-all-ones term (CANONICAL_INSTRUCTION_ADDRESS).
 * jolt_host_lookup_prefix_evaluate / jolt_host_lookup_table_combine expose the two building blocks for the decomposition test
 * (tables/test_utils.rs prefix_suffix_test): b of b_len <= 16 bits (even), checkpoints / prefixes = all 49 slots. */
typedef struct jolt_read_raf_address jolt_read_raf_address;
uint32_t jolt_lookup_table_count(void);
uint32_t jolt_lookup_prefix_count(void);
int32_t jolt_lookup_table_suffixes(uint32_t kind, uint8_t *kinds_out /* <= 5 */, uint32_t *n_out);
int32_t jolt_lookup_table_prefixes(uint32_t kind, uint8_t *prefixes_out /* <= 4 */, uint32_t *n_out);
int32_t jolt_lookup_suffix_layout(uint32_t *offsets_out /* 43 */, uint8_t *kinds_out /* offsets_out[42]; may be NULL */);
int32_t jolt_host_lookup_prefix_default_checkpoints(jolt_fr_t *out /* 49 */);
int32_t jolt_host_lookup_prefix_evaluate(uint32_t prefix, const jolt_fr_t *checkpoints, uint32_t b, uint32_t b_len, uint32_t suffix_len, jolt_fr_t *out);
int32_t jolt_host_lookup_prefix_table(uint32_t prefix, const jolt_fr_t *checkpoints, uint32_t b_len, uint32_t suffix_len, jolt_fr_t *out /* 2^b_len */);
int32_t jolt_host_lookup_table_combine(uint32_t kind, const jolt_fr_t *prefixes, const jolt_fr_t *suffixes, jolt_fr_t *out);
int32_t jolt_host_read_raf_address_create(const jolt_fr_t *gamma, const uint8_t *table_present /* 42 */, int32_t canonical, jolt_read_raf_address **out);
int32_t jolt_host_read_raf_address_destroy(jolt_read_raf_address *h);
int32_t jolt_host_read_raf_address_init_phase(jolt_read_raf_address *h, uint32_t phase, const jolt_fr_t *raf_sums, const jolt_fr_t *suffix_sums);
int32_t jolt_host_read_raf_address_message(jolt_read_raf_address *h, const jolt_fr_t *previous_claim, jolt_fr_t *evals_out /* 3 */);
int32_t jolt_host_read_raf_address_bind(jolt_read_raf_address *h, const jolt_fr_t *challenge, int32_t *phase_done);
/* bind + the next round's message in one hand-off (ProveRounds::prove_round with Some(bind), prover.rs:45-51), for the 1st .. 7th bind of a phase */
int32_t jolt_host_read_raf_address_bind_message(jolt_read_raf_address *h, const jolt_fr_t *challenge, const jolt_fr_t *previous_claim, jolt_fr_t *evals_out /* 3 */);
int32_t jolt_host_read_raf_address_prove_phase(jolt_read_raf_address *h, jolt_fr_t *claim, jolt_round_transcript_fn fn, void *user, jolt_host_transcript *test_transcript,
                                               jolt_fr_t *coeffs_out /* 8 x 3; may be NULL */, jolt_fr_t *challenges_out /* 8; may be NULL */);
int32_t jolt_host_read_raf_address_v_table(const jolt_read_raf_address *h, uint32_t phase, jolt_fr_t *out /* 256 */);
int32_t jolt_host_read_raf_address_finish(const jolt_read_raf_address *h, jolt_fr_t *table_values /* 42 */, jolt_fr_t *raf_interleaved, jolt_fr_t *raf_identity);

/* Spartan outer (stage 1) T-scale sums -- SURVEY.md section 8(f) row 3 (crates/jolt-kernels/src/{reference,optimized}/spartan_outer.rs).
 * The constraint list, spartan_outer_row_weights and the Lagrange interpolation stay in Rust (O(rows) work); the caller folds the
 * per-(node, stream) row weights into per-column weights (ConstraintMatrices::weighted_columns + public_column_contributions,
 * reference/spartan_outer.rs:246-256): weights are laid out [node][stream][1 + n_inputs], column 0 = the constant.
 *   jolt_r1cs_uniskip_sums: out[node] = sum_t sum_s eq[(t << 1) | s] * Az(node,s,t) * Bz(node,s,t), Az(node,s,t) = c_0 + sum_v c_v z_v(t)
 *     -- the extended-node evaluations t1 of uniskip_first_round_poly (:172-221); eq has 2 * cycles entries (tau_low, stream = LSB).
 *   jolt_r1cs_materialize: az[(t << 1) | s], bz[(t << 1) | s] for ONE (node =) uni-skip challenge: the remainder member's linear
 *     forms over the joint (cycle || stream) domain (:236-300); feed them to jolt_member_create_split_eq_product with w = tau_low
 *     and scale = the Lagrange kernel value for all log_t + 1 remainder rounds.
 *   jolt_tables_evaluate: out[k] = tables[k] evaluated at `point`, all from one eq expansion (the post-hoc opening evaluation of
 *     the 35 inputs, optimized/spartan_outer.rs:41-43; Polynomial::evaluate, dense.rs:340-366).  <= 64 tables. */
int32_t jolt_r1cs_uniskip_sums(jolt_ctx *ctx, jolt_table *const *inputs, size_t n_inputs, const jolt_table *eq, const jolt_fr_t *a_weights,
                               const jolt_fr_t *b_weights, size_t n_nodes, jolt_fr_t *out);
int32_t jolt_r1cs_materialize(jolt_ctx *ctx, jolt_table *const *inputs, size_t n_inputs, const jolt_fr_t *a_weights, const jolt_fr_t *b_weights,
                              jolt_table **az_out, jolt_table **bz_out);
int32_t jolt_tables_evaluate(jolt_ctx *ctx, jolt_table *const *tables, size_t k, const jolt_fr_t *point, size_t n, jolt_fr_t *out);

/* The same three operators straight off the INTEGER witness columns (jolt_ints of kind u64 / i64 / i128), as the optimized tier runs
 * them (crates/jolt-kernels/src/optimized/spartan_outer.rs:6-43,276-370,780-850) on FrSmallScalarAccumulator-style deferred reduction
 * (crates/jolt-field/src/bn254/mont.rs:286-305,343-427): the 35 inputs are never promoted to field tables.
 *   jolt_r1cs_uniskip_sums_small: INTEGER weights (the exact integer Lagrange extension coefficients folded into column weights,
 *     extension_coefficients :276-306): Az / Bz are integer dot products, their product times eq is one field multiply per
 *     (node, cycle, stream).  Contract (the caller's, as in the reference: |az| < 2^22, |bz| < 2^152 there): |Az| < 2^127,
 *     |Bz| < 2^255, |Az * Bz| < 2^254, no weight equal to INT64_MIN.
 *   jolt_r1cs_materialize_small: FIELD weights (they carry the Lagrange kernel at the uni-skip challenge) x integer values, one
 *     reduction per output (fold_group :363-370).
 *   jolt_ints_evaluate: out[k] = sum_t eq(point, t) * z_k(t) (compute_claimed_inputs :780-850; Polynomial::<T>::evaluate).
 * n_streams = 2: Spartan outer (weights [node][stream][1 + n], eq over (cycle || stream), outputs az[(t << 1) | s]).
 * n_streams = 1: Spartan PRODUCT virtualization (stage 2, crates/jolt-kernels/src/optimized/spartan_product.rs:86-230,321-437): "A" = the
 *   left factor lanes, "B" = the right factor lanes, 5 extended nodes of the 3-node window, eq = eq(tau_low, .) over the cycles, and the
 *   two outputs of jolt_r1cs_materialize_small are the remainder's left / right tables (feed jolt_member_create_split_eq_product).
 * Results equal the field-arithmetic operators above on the promoted columns (exact algebra). */
int32_t jolt_r1cs_uniskip_sums_small(jolt_ctx *ctx, const jolt_ints *const *inputs, size_t n_inputs, const jolt_table *eq, uint32_t n_streams,
                                     const int64_t *a_weights, const int64_t *b_weights, size_t n_nodes, jolt_fr_t *out);
int32_t jolt_r1cs_materialize_small(jolt_ctx *ctx, const jolt_ints *const *inputs, size_t n_inputs, uint32_t n_streams, const jolt_fr_t *a_weights,
                                    const jolt_fr_t *b_weights, jolt_table **az_out, jolt_table **bz_out);
int32_t jolt_ints_evaluate(jolt_ctx *ctx, const jolt_ints *const *columns, size_t k, const jolt_fr_t *point, size_t n, jolt_fr_t *out);
/* csrc/small_scalar.hip.h built for the host (CPU suite): sum_k values[k] * scalars[k], scalars as signed 128-bit (lo, hi) pairs. */
int32_t jolt_host_small_scalar_dot(const jolt_fr_t *values, const uint64_t *scalars, size_t n, jolt_fr_t *out);

/* Sparse (K x T) read-write matrix of RAM read/write checking (stage 2) -- SURVEY.md section 8(f) row 4.  Replaces
 * CycleMajorMatrix / AddressMajorMatrix and the round messages of RamReadWriteKernel (crates/jolt-kernels/src/optimized/rw_matrix.rs,
 * optimized/ram_read_write.rs:58-330): summand eq(tau_low, j) * ra(k,j) * (val(k,j) + gamma * (val(k,j) + inc(j))) over
 * (address || cycle), bound low-to-high with the log_t cycle variables first (the default read-write config, the only one the
 * reference kernels support, ram_read_write.rs:281-286); one entry per RAM access, never a K x T grid.
 *   create: addresses[j] = remapped word address of cycle j or UINT64_MAX (RamAccessColumns, optimized/ram_trace.rs:22-75),
 *     pre / post = the word before / after the access; inc (T entries) and val_init (K entries) are copied.
 *   prove_round: ProveRounds::prove_round, device half.  Rounds < log_t return (q(0), q_inf) of the quadratic factor and
 *     aux_out = {current_scalar, tau_low[current_index - 1], 0}: the caller completes the cubic with gruen_poly_deg_3
 *     (split_eq.rs:383-417); later rounds return (s(0), s(2)), s(1) comes from the claim (UnivariatePoly::from_evals_and_hint).
 *   final_values: {ra, val, inc, bound cycle-eq factor} (RamReadWriteOutputClaims + validate_derived_tables). */
typedef struct jolt_rw_matrix jolt_rw_matrix;
int32_t jolt_rw_matrix_create(jolt_ctx *ctx, const uint64_t *addresses, const uint64_t *pre_values, const uint64_t *post_values, size_t cycles,
                              const jolt_table *inc, const jolt_table *val_init, const jolt_fr_t *tau_low, const jolt_fr_t *gamma,
                              jolt_rw_matrix **out);
/* The same member over access columns already resident in HBM (three JOLT_INT_U64 jolt_ints of `cycles` entries, uploaded once per trace):
 * no host pass over the columns and no upload per proof. */
int32_t jolt_rw_matrix_create_resident(jolt_ctx *ctx, const jolt_ints *addresses, const jolt_ints *pre_values, const jolt_ints *post_values,
                                       const jolt_table *inc, const jolt_table *val_init, const jolt_fr_t *tau_low, const jolt_fr_t *gamma,
                                       jolt_rw_matrix **out);

/* Registers read/write checking (stage 4) -- SURVEY.md section 8(f) row 4.  Replaces the sparse cycle-major matrix and the round messages of
 * OptimizedRegistersReadWrite (crates/jolt-kernels/src/optimized/registers_read_write/{mod.rs:79-402, sparse.rs, rows.rs}): summand
 *   eq(r_cycle, j) * ( rd_wa * (rd_inc + val) + gamma * rs1_ra * val + gamma^2 * rs2_ra * val )(k, j)   (reference/registers_read_write.rs:3-10)
 * over (register || cycle), bound low-to-high, the log T cycle variables first; <= 3 cells per cycle (RegisterCycleRow::entries), the
 * gamma-combined read coefficient and the write coefficient as two columns of the same sparse matrix as jolt_rw_matrix, K-sized dense
 * arrays on the host for the log K address rounds.  All inputs are resident: `regs` holds the hot-index columns rs1, rs2, rd (polys 0, 1, 2;
 * k = 2^REGISTER_ADDRESS_BITS), the value columns are JOLT_INT_U64 jolt_ints, `inc` is RdInc (copied).
 *   prove_round: cycle rounds return (q(0), leading coefficient) in evals_out[0..1] and aux_out = {current_scalar, r_cycle[current_index - 1], 0}
 *     (the caller completes the cubic with gruen_poly_deg_3); address rounds return s(0), s(1), s(2), s(3) in evals_out[0..3]
 *     (UnivariatePoly::from_evals).  finish / destroy / len: jolt_rw_matrix_finish / _destroy / _len.
 *   final_values: {registers_val, rd_wa, gamma * rs1_ra + gamma^2 * rs2_ra, rd_inc, bound cycle-eq factor}; the operand claims rs1_ra / rs2_ra
 *     are evaluations of the index columns at the bound point: jolt_onehot_materialize(regs, p, eq(r_address, .)) then jolt_evaluate at r_cycle. */
int32_t jolt_registers_rw_create(jolt_ctx *ctx, const jolt_onehot *regs, const jolt_ints *rs1_val, const jolt_ints *rs2_val, const jolt_ints *rd_pre,
                                 const jolt_ints *rd_post, const jolt_table *inc, const jolt_fr_t *r_cycle, const jolt_fr_t *gamma, jolt_rw_matrix **out);
int32_t jolt_registers_rw_prove_round(jolt_rw_matrix *m, const jolt_fr_t *bind, jolt_fr_t *evals_out /* 4 */, jolt_fr_t *aux_out /* 3, may be NULL */);
int32_t jolt_registers_rw_final_values(jolt_rw_matrix *m, jolt_fr_t *out /* 5 */);
/* test hook: the cells of the cycle phase */
int32_t jolt_registers_rw_download(jolt_rw_matrix *m, uint64_t *rows, uint64_t *cols, jolt_fr_t *val, jolt_fr_t *ra, jolt_fr_t *wa, uint64_t *prev, uint64_t *next);
int32_t jolt_rw_matrix_prove_round(jolt_rw_matrix *m, const jolt_fr_t *bind, jolt_fr_t *evals_out /* 2 */, jolt_fr_t *aux_out /* 3 or NULL */);
int32_t jolt_rw_matrix_finish(jolt_rw_matrix *m, const jolt_fr_t *bind);
int32_t jolt_rw_matrix_final_values(jolt_rw_matrix *m, jolt_fr_t *out /* 4 */);

/* Pushforwards of cycle weights onto a LARGE address domain (bytecode PCs, RAM words; K up to 2^24): the T-scale half of the joint-domain
 * relations whose rounds run over K-sized tables.  Replaces stage_pushforwards of the bytecode read+RAF address phase
 * (crates/jolt-kernels/src/optimized/bytecode_read_raf.rs:152-237: F_s(k) = sum_{j: pc(j) = k} eq(r_cycle_s, j) for the five stages in one
 * trace walk) and RamAccessColumns::fold_cycles (optimized/ram_trace.rs:150-162, used by optimized/ram_raf_evaluation.rs:44-48).
 *   jolt_key_index_create: sorts the rows of a resident key column (JOLT_INT_U64, one key per cycle; a key >= K is a cold cycle and
 *     contributes nothing: NO_ACCESS, unmapped PCs) by key, once per trace column -- the reference shares the same scan through its
 *     ProofSession (PcRow::shared :92-150, RamAccessColumns::shared).
 *   jolt_key_index_pushforward: out[s][k] = sum_{j: key(j) = k} weights[s][j] for n_weights <= 8 tables of `cycles` entries at once; new
 *     tables of K entries (the caller frees them).
 *   jolt_key_index_last_value: out[k] = values[latest cycle with key k] as a field element, init[k] for a key that never occurs: the final
 *     state of a memory that is only ever overwritten (the ram_val_final column of the RAM output check, optimized/ram_output_check.rs:91). */
typedef struct jolt_key_index jolt_key_index;
int32_t jolt_key_index_create(jolt_ctx *ctx, const jolt_ints *keys, uint64_t k, jolt_key_index **out);
int32_t jolt_key_index_size(const jolt_key_index *index, size_t *cycles, uint64_t *k, uint32_t *items);
int32_t jolt_key_index_pushforward(jolt_ctx *ctx, const jolt_key_index *index, jolt_table *const *weights, size_t n_weights, jolt_table **out);
int32_t jolt_key_index_last_value(jolt_ctx *ctx, const jolt_key_index *index, const jolt_ints *values, const jolt_table *init, jolt_table **out);
int32_t jolt_key_index_destroy(jolt_ctx *ctx, jolt_key_index *index);

/* The hypercube-sharded form of both matrices (one process per GPU; the reference has no multi-GPU code: this is north_star's "the 2^n boolean hypercube shards
 * across the GPUs" for rw_matrix.rs / registers_read_write/sparse.rs).  Cycles are dealt to the ranks in contiguous blocks and bind low to high, so rank g runs the
 * first log T_local cycle rounds on a LOCAL matrix over its block (the constructors above with the low log T_local coordinates of the cycle point; its round sums,
 * scaled by eq(w_hi, g), add over the ranks).  jolt_rw_matrix_hold_row (before the rounds) keeps the single row those rounds leave in cycle-major form;
 * jolt_rw_matrix_bind ingests the last local challenge; jolt_rw_matrix_export_row reads the row out (cells in column order, raw checkpoints, the bound increment, the
 * bound eq factor).  The rows of all ranks stacked in rank order are the matrix of the remaining log G cycle variables: jolt_rw_matrix_create_merged builds it on
 * every rank (w = the log_rows HIGH coordinates, inc = the ranks' bound increments in rank order) and the ordinary prove_round / finish / final_values continue. */
int32_t jolt_rw_matrix_hold_row(jolt_rw_matrix *m);
int32_t jolt_rw_matrix_bind(jolt_rw_matrix *m, const jolt_fr_t *bind);
int32_t jolt_rw_matrix_export_row(jolt_rw_matrix *m, size_t cap, uint64_t *cols, uint64_t *prev, uint64_t *next, jolt_fr_t *val, jolt_fr_t *ra, jolt_fr_t *wa /* registers */,
                                  jolt_fr_t *inc_out, jolt_fr_t *scalar_out, size_t *n_out);
int32_t jolt_rw_matrix_create_merged(jolt_ctx *ctx, int32_t registers, size_t log_rows, size_t log_k, size_t n, const uint64_t *rows, const uint64_t *cols,
                                     const uint64_t *prev, const uint64_t *next, const jolt_fr_t *val, const jolt_fr_t *ra, const jolt_fr_t *wa, const jolt_fr_t *inc,
                                     const jolt_table *val_init, const jolt_fr_t *w, const jolt_fr_t *scalar, const jolt_fr_t *gamma, jolt_rw_matrix **out);
int32_t jolt_rw_matrix_len(const jolt_rw_matrix *m, size_t *entries);
int32_t jolt_rw_matrix_download(jolt_rw_matrix *m, uint64_t *rows, uint64_t *cols, jolt_fr_t *val, jolt_fr_t *ra, jolt_fr_t *prev, jolt_fr_t *next);
int32_t jolt_rw_matrix_destroy(jolt_rw_matrix *m);

/* Multi-GPU data path (one process per GPU).  The collectives are RCCL calls on the context's stream;
 * librccl.so.1 is resolved with dlopen at first use (`rccl_path` may name it explicitly, NULL = the copy already mapped into
 * the process / the default search path).  Rank 0 draws the id, the launcher broadcasts its 128 bytes (torch.distributed
 * in this repo, any side channel in the Rust host), every rank then calls jolt_comm_create (collective).
 * Replaces nothing in the reference -- the reference prover is single-process. */
typedef struct jolt_comm jolt_comm;
int32_t jolt_comm_unique_id(const char *rccl_path, uint8_t out[128]);
int32_t jolt_comm_create(jolt_ctx *ctx, const char *rccl_path, const uint8_t unique_id[128], int32_t rank, int32_t world, jolt_comm **out);
int32_t jolt_comm_destroy(jolt_comm *comm);
int32_t jolt_comm_world(const jolt_comm *comm, int32_t *rank, int32_t *world);
/* gathered = world blocks of `bytes` in rank order.  _host: small host payload (per-round partial sums), synchronous;
 * _device / _table: device buffers, asynchronous on the context stream. */
int32_t jolt_comm_all_gather_host(jolt_comm *comm, const void *local, size_t bytes, void *gathered);
int32_t jolt_comm_all_gather_device(jolt_comm *comm, const void *d_local, size_t bytes, void *d_gathered);
int32_t jolt_comm_all_gather_table(jolt_comm *comm, const jolt_table *local, size_t n, jolt_table *gathered);
/* A jolt_gather_fn for jolt_host_batch_run with user = jolt_comm*. */
int32_t jolt_comm_gather_round_sums(void *user, const jolt_fr_t *local, size_t count, jolt_fr_t *gathered);
/* The same exchange between the ranks of ONE node through POSIX shared memory (host memory only; ~1 us instead of an RCCL
 * all-gather on the critical path of every sharded round).  Collective: rank 0 creates the segment `name`
 * ("/..."), the others attach; JOLT_ERR_UNSUPPORTED when the segment cannot be created / mapped (the launcher then keeps the
 * RCCL exchange on every rank).  max_bytes bounds one rank's payload. */
typedef struct jolt_shm jolt_shm;
int32_t jolt_shm_create(const char *name, int32_t rank, int32_t world, size_t max_bytes, jolt_shm **out);
/* The same with a per-run nonce (non-zero; the launcher draws it on rank 0 and distributes it with the name): an attaching rank
 * accepts only the segment rank 0 initialised with this value, never a stale one a crashed run left under the same name. */
int32_t jolt_shm_create_nonce(const char *name, uint64_t nonce, int32_t rank, int32_t world, size_t max_bytes, jolt_shm **out);
int32_t jolt_shm_destroy(jolt_shm *shm);
int32_t jolt_shm_all_gather(jolt_shm *shm, const void *local, size_t bytes, void *gathered);
/* A jolt_gather_fn for jolt_host_batch_run with user = jolt_shm*. */
int32_t jolt_shm_gather_round_sums(void *user, const jolt_fr_t *local, size_t count, jolt_fr_t *gathered);
/* Hand-over to the redundant tail: pack the current tables (`entries` values each) of several members table-major into dst;
 * after the all-gather, dst[t][r*entries + j] = gathered[r][t][j] (rank = top variables). */
int32_t jolt_round_group_pack_tables(jolt_ctx *ctx, jolt_member *const *members, size_t n_members, size_t entries, jolt_table *dst);
int32_t jolt_tail_interleave(jolt_ctx *ctx, const jolt_table *gathered, size_t world, size_t n_tables, size_t entries, jolt_table *dst);
/* The fully bound values of several members with one copy and one synchronisation (concatenated in member order). */
int32_t jolt_round_group_final_values(jolt_ctx *ctx, jolt_member *const *members, size_t n_members, jolt_fr_t *out, size_t capacity);

/* HyperKZGScheme::commit / open (crates/jolt-hyperkzg/src/scheme.rs:122-158,302-325, kzg.rs:69-126) over the device
 * kernels with the same test transcript.  com: ell-1 points, w: 3 points, v: 3*ell, challenges: {r, q, d_0}. */
int32_t jolt_host_hyperkzg_commit(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *evals, jolt_g1_t *out);
int32_t jolt_host_hyperkzg_open(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *evals, const jolt_fr_t *point, size_t ell,
                                uint64_t transcript_label, jolt_g1_t *com, jolt_g1_t *w, jolt_fr_t *v, jolt_fr_t *challenges_out);

/* The same opening under the CALLER's Fiat-Shamir transcript -- CommitmentScheme::open(poly, point, eval, setup, hint, transcript) (crates/jolt-openings/src/
 * schemes.rs:66-72; impl crates/jolt-hyperkzg/src/scheme.rs:314-325) with the polynomial resident in HBM.  `fn` is called three times per opening, in order:
 *   phase 0: points = the ell - 1 level commitments (scheme.rs:148-152)        -> challenge r
 *   phase 1: values = v[t][j], 3 * ell field elements row-major (kzg.rs:88-95) -> challenge q
 *   phase 2: points = the three witness commitments (kzg.rs:118-124)            -> challenge d_0
 * it absorbs them (transcript.append per element, in order) and returns the challenge; a non-zero return aborts the opening with that status. */
typedef int32_t (*jolt_open_transcript_fn)(void *user, int32_t phase, const jolt_g1_t *points, size_t n_points, const jolt_fr_t *values, size_t n_values,
                                           jolt_fr_t *challenge_out);
int32_t jolt_host_hyperkzg_open_with_transcript(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *evals, const jolt_fr_t *point, size_t ell,
                                                jolt_open_transcript_fn fn, void *user, jolt_g1_t *com, jolt_g1_t *w, jolt_fr_t *v, jolt_fr_t *challenges_out);
/* ... and with its first n_known level commitments supplied by the caller (HyperKZGScheme::open commits every folded polynomial by MSM, scheme.rs:141-145; for the
 * joint polynomial of one-hot and dense columns the first folds' commitments follow by linearity from jolt_grid_commit_onehot_classes): known_levels[i] is absorbed and
 * returned as com[i], the MSMs of those levels are skipped.  fn == NULL: the library's test transcript with transcript_label.  Single-process opening only. */
int32_t jolt_host_hyperkzg_open_with_levels(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *evals, const jolt_fr_t *point, size_t ell, uint64_t transcript_label,
                                            jolt_open_transcript_fn fn, void *user, const jolt_g1_t *known_levels, size_t n_known, jolt_g1_t *com, jolt_g1_t *w,
                                            jolt_fr_t *v, jolt_fr_t *challenges_out);

/* Term-range pieces of a commitment / opening sharded over the ranks of a node (the reference is single
 * process): an MSM of n terms against srs[base_offset .. base_offset + n) with the scalars scalars[scalar_offset ..]; the partial
 * one-hot commitments over cycles [cycle_lo, cycle_hi); and HyperKZGScheme::open with EVERY MSM split by term range over `world`
 * ranks -- each rank holds the polynomial and the bases, multiplies terms [n*rank/world, n*(rank+1)/world), the partial points are
 * all-gathered through `gather` (a jolt_gather_fn moving 32-byte words: jolt_comm_gather_round_sums, jolt_shm_gather_round_sums)
 * and added in rank order, so every rank absorbs the same commitments and returns the same proof as jolt_host_hyperkzg_open. */
int32_t jolt_msm_g1_table_range(jolt_ctx *ctx, const jolt_srs *srs, size_t base_offset, const jolt_table *scalars, size_t scalar_offset,
                                size_t n, jolt_g1_t *out);
/* One window of a STREAMED commitment: StreamingCommitment::{feed, feed_u64, feed_i128, feed_i128_rows_with} for a KZG-type scheme
 * (crates/jolt-openings/src/schemes.rs:167-222; the callers: crates/jolt-kernels/src/reference/commitment.rs:86-121, optimized/commitment.rs:317-352).  The
 * polynomial arrives in coefficient order as host slices; out = acc + sum_{i<n} values[i] * srs[base_offset + i] (acc == NULL: the identity), values being n field
 * elements (kind JOLT_SCALAR_FR) or machine integers promoted by Ring::from_u64 / from_i64 / from_i128 (kind JOLT_INT_*).  Stateless: the caller's PartialCommitment is
 * the pair (point, next coefficient index).  base_offset + n beyond the SRS is JOLT_ERR_SRS_TOO_SMALL. */
enum { JOLT_SCALAR_FR = 3 };
int32_t jolt_msm_g1_window(jolt_ctx *ctx, const jolt_srs *srs, size_t base_offset, int32_t kind, const void *host, size_t n, const jolt_g1_t *acc, jolt_g1_t *out);
int32_t jolt_grid_commit_onehot_range(jolt_ctx *ctx, const jolt_srs *srs, const jolt_onehot *source, size_t cycle_lo, size_t cycle_hi,
                                      jolt_g1_t *out /* n_polys */);
/* out[c * n_polys + p] = sum over the cycles j = c (mod 2^shift) of srs[(hot_p(j) * T + j) >> shift]: the commitments of the 2^shift residue-class parts of every column
 * on the grid of a polynomial folded `shift` times low to high (shift <= 4, T a power of two).  com(fold of the joint polynomial) is a linear combination of them:
 * for shift = 1 and the fold variable x, sum_p s_p ((1 - x) out[0][p] + x out[1][p]) + com(the dense columns' fold). */
int32_t jolt_grid_commit_onehot_classes(jolt_ctx *ctx, const jolt_srs *srs, const jolt_onehot *source, uint32_t shift, jolt_g1_t *out);
int32_t jolt_host_hyperkzg_open_sharded(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *evals, const jolt_fr_t *point, size_t ell,
                                        uint64_t transcript_label, int32_t rank, int32_t world, jolt_gather_fn gather, void *user,
                                        jolt_g1_t *com, jolt_g1_t *w, jolt_fr_t *v, jolt_fr_t *challenges_out);
/* The BLOCK-CYCLIC term assignment, the one that keeps the fixed-base window tables at any world size: term i
 * of every MSM belongs to rank (i / block) % world, and a rank's jolt_srs holds exactly the bases of its terms, compacted in index
 * order (count_global / world points; upload them with jolt_srs_upload_g1, or generate them from a test secret here).  The terms a rank
 * owns of any prefix [0, n) are then a prefix of its compact SRS, so jolt_srs_precompute_windows over the compact SRS serves every
 * level of an opening.  With block = the rank's cycle count T, a rank's compact SRS is the commitment grid of ITS cycles (position
 * k*T + j), so its share of a commitment is jolt_grid_commit_onehot / jolt_msm_g1_table_range over local columns.
 * jolt_msm_g1_table_blocks: the rank's share of sum_{i<n} scalars[i]*SRS[i] (the shares of all ranks add up to the MSM);
 * jolt_host_hyperkzg_open_sharded_blocks: jolt_host_hyperkzg_open_sharded with this assignment. */
/* how many of the terms [0, n) rank owns = the length of its compact prefix (host only, no GPU) */
int32_t jolt_host_owned_terms(size_t n, size_t block, int32_t rank, int32_t world, size_t *out);
int32_t jolt_srs_setup_from_secret_blocks(jolt_ctx *ctx, const jolt_fr_t *beta, size_t count_global, const jolt_g1_t *g1, size_t block,
                                          int32_t rank, int32_t world, jolt_srs **out);
int32_t jolt_msm_g1_table_blocks(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *scalars, size_t n, size_t block, int32_t rank,
                                 int32_t world, jolt_g1_t *out);
int32_t jolt_host_hyperkzg_open_sharded_blocks(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *evals, const jolt_fr_t *point,
                                               size_t ell, uint64_t transcript_label, int32_t rank, int32_t world, size_t block,
                                               jolt_gather_fn gather, void *user, jolt_g1_t *com, jolt_g1_t *w, jolt_fr_t *v,
                                               jolt_fr_t *challenges_out);
/* The SUBTREE assignment (world = 2^gamma ranks): the gamma bits below the leading one of an index name its owner
 * (indices below `world`: owner = the index, slot 0; otherwise slot = the index with those bits removed).  Like the block-cyclic one it
 * keeps every prefix of the indices a prefix of every rank's compact arrays -- one compact SRS and one set of window tables per rank
 * -- and it is also closed under LowToHigh folding (slots 2c, 2c+1 of a level fold into slot c of the next), so the POLYNOMIAL of an
 * opening is sharded too: jolt_host_hyperkzg_open_subtree takes the rank's compact array of the evaluations (2^ell / world
 * coefficients: jolt_grid_joint_polynomial_subtree builds it for the commitment grid; jolt_host_subtree_term_index gives the index
 * of a slot for any other source) and the rank's compact SRS, runs folds / RLC / Horner passes / quotient scans / MSMs on 1 / world
 * of the data, exchanges O(ell) field elements and O(ell) points through `gather`, and returns on every rank the proof
 * jolt_host_hyperkzg_open returns for the whole polynomial (world >= 2, ell > log2 world; one rank: jolt_host_hyperkzg_open).
 * tests/subtree_model.py is the executable specification. */
int32_t jolt_host_subtree_owned_terms(size_t n, int32_t rank, int32_t world, size_t *out);
int32_t jolt_host_subtree_term_index(size_t slot, int32_t rank, int32_t world, size_t *out);
int32_t jolt_srs_setup_from_secret_subtree(jolt_ctx *ctx, const jolt_fr_t *beta, size_t count_global, const jolt_g1_t *g1, int32_t rank,
                                           int32_t world, jolt_srs **out);
int32_t jolt_msm_g1_table_subtree(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *scalars, size_t n, int32_t rank, int32_t world,
                                  jolt_g1_t *out);
int32_t jolt_grid_joint_polynomial_subtree(jolt_ctx *ctx, const jolt_onehot *const *sources, size_t n_sources,
                                           const jolt_fr_t *onehot_scalars, jolt_table *const *dense, size_t n_dense,
                                           const jolt_fr_t *dense_scalars, uint32_t log_k, int32_t rank, int32_t world, jolt_table **out);
int32_t jolt_host_hyperkzg_open_subtree(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *evals, const jolt_fr_t *point, size_t ell,
                                        uint64_t transcript_label, int32_t rank, int32_t world, jolt_gather_fn gather, void *user,
                                        jolt_g1_t *com, jolt_g1_t *w, jolt_fr_t *v, jolt_fr_t *challenges_out);

/* The commitment grid's OPENING HINT -- CommitmentScheme::commit returns (Commitment, OpeningHint) (crates/jolt-openings/src/schemes.rs:60-72): what the committer can
 * precompute for the opening from the polynomials and the setup alone.  For the grid's one-hot columns: per fold depth s = 1 .. levels (<= 4) the residue-class sums
 * S_p^(s, c) = sum over the cycles j = c mod 2^s of srs[(hot_p(j) * T + j) >> s] (what jolt_grid_commit_onehot_classes returns to the host), kept on the device.
 * None of it depends on a challenge.  background != 0: enqueued on the context's lowest-priority stream at one wavefront per SIMD and returned at once -- the sums run
 * under the latency-bound legs between the commitment and the opening; 0: on the main stream.  jolt_host_hyperkzg_open_grid turns them into the opening's first level
 * commitments by linearity (same points as scheme.rs:141-145 commits by MSM).  The sources must outlive the hint's sums (jolt_grid_hint_wait, or the opening). */
typedef struct jolt_grid_hint jolt_grid_hint;
int32_t jolt_grid_hint_begin(jolt_ctx *ctx, const jolt_srs *srs, const jolt_onehot *const *sources, size_t n_sources, uint32_t levels, int32_t background,
                             jolt_grid_hint **out);
int32_t jolt_grid_hint_wait(jolt_ctx *ctx, jolt_grid_hint *hint);
int32_t jolt_grid_hint_download(jolt_ctx *ctx, jolt_grid_hint *hint, uint32_t level, jolt_g1_t *out /* 2^level x n_cols, [class][column] */); /* test hook */
int32_t jolt_grid_hint_free(jolt_ctx *ctx, jolt_grid_hint *hint);
/* HyperKZGScheme::open (crates/jolt-hyperkzg/src/scheme.rs:122-158) of the grid's joint polynomial `evals` (jolt_grid_joint_polynomial over the same sources, scalars and
 * dense columns) with the first `levels` level commitments by linearity: one-hot part from the hint, dense part as (T >> s)-term MSMs of the dense columns' folds inside
 * the same MSM pipeline as the remaining levels.  fn == NULL: the library's test transcript under transcript_label.  Proof identical to jolt_host_hyperkzg_open(evals). */
int32_t jolt_host_hyperkzg_open_grid(jolt_ctx *ctx, const jolt_srs *srs, const jolt_table *evals, const jolt_fr_t *point, size_t ell, uint64_t transcript_label,
                                     jolt_open_transcript_fn fn, void *user, const jolt_grid_hint *hint, uint32_t levels, const jolt_fr_t *onehot_scalars,
                                     jolt_table *const *dense, size_t n_dense, const jolt_fr_t *dense_scalars, jolt_g1_t *com, jolt_g1_t *w, jolt_fr_t *v,
                                     jolt_fr_t *challenges_out);

/* ------------------------------------------------------------------------------------------------------------
 * Stage operators as ProveRounds objects -- one per backend slot (crates/jolt-kernels/src/backend.rs:126-171).
 *
 * A slot is a PrepareKernel<F, R> whose prepare(..) returns Box<dyn SumcheckKernel<F, Relation = R>> (backend.rs:98-111, kernel.rs:72-126):
 * ProveRounds (crates/jolt-sumcheck/src/prover.rs:52-72) + output_claims.  jolt_stage_<operator>_create IS that prepare: every T-scale pass that
 * depends on no round challenge runs there, over inputs that are resident in HBM (borrowed: they must outlive the operator).  The object then
 * follows the fused contract (prover.rs:45-51): prove_round(bind = the PREVIOUS round's challenge or NULL in the first active round, round,
 * previous_claim) returns the round message as UnivariatePoly coefficients c_0 .. c_{n-1} (n <= degree + 1; s(0) + s(1) = previous_claim),
 * finish_rounds(bind) applies the last challenge, output_claims are SumcheckKernel::output_claims in the order each constructor documents.
 * Errors as everywhere in this header; a message that fails its own round check is JOLT_ERR_ROUND_CHECK.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct jolt_stage_op jolt_stage_op;
int32_t jolt_stage_op_num_rounds(const jolt_stage_op *op, size_t *rounds);                  /* ProveRounds::num_rounds */
int32_t jolt_stage_op_degree(const jolt_stage_op *op, size_t *degree);                      /* the largest degree of a round message */
int32_t jolt_stage_op_input_claim(jolt_stage_op *op, jolt_fr_t *claim);                     /* test / bench convenience: a prover holds it from the earlier stages */
int32_t jolt_stage_op_prove_round(jolt_stage_op *op, const jolt_fr_t *bind, size_t round, const jolt_fr_t *previous_claim, jolt_fr_t *coeffs_out, size_t cap,
                                  size_t *n_coeffs);                                        /* ProveRounds::prove_round (prover.rs:57-66) */
int32_t jolt_stage_op_finish_rounds(jolt_stage_op *op, const jolt_fr_t *bind);              /* ProveRounds::finish_rounds (prover.rs:68-71) */
int32_t jolt_stage_op_output_claims(jolt_stage_op *op, jolt_fr_t *out, size_t cap, size_t *n); /* SumcheckKernel::output_claims (kernel.rs:86-92) */
/* Intermediate values an operator keeps on the host for parity tests, by name: "masses" (pushforwards: n_polys x K), "scan_raf" / "scan_suffix" (the 16 read-RAF
 * phase scans), "v_tables", "table_values", "raf_values", "cycle_claim".  out == NULL asks for the count. */
int32_t jolt_stage_op_kept(const jolt_stage_op *op, const char *key, jolt_fr_t *out, size_t cap, size_t *n);
/* Rounds [first, first + n) of `parent` as an operator of its own (the parent stays alive and owns the state): a caller can put the phases of one kernel under
 * different transcripts or drivers; a window's last challenge is handed to the parent with the next window's first round. */
int32_t jolt_stage_op_window(jolt_stage_op *parent, size_t first, size_t n, jolt_stage_op **out);
int32_t jolt_stage_op_destroy(jolt_stage_op *op);

/* UniskipKernel (crates/jolt-kernels/src/uniskip.rs:28-54) of Spartan outer (n_streams = 2) / product (n_streams = 1): the extended-node sums t1 of the uni-skip
 * first round off the integer columns (jolt_r1cs_uniskip_sums_small) against eq(tau, .) expanded here.  tau: n_tau coordinates (cycle variables, then the stream). */
int32_t jolt_stage_spartan_uniskip_sums(jolt_ctx *ctx, const jolt_ints *const *cols, size_t n_cols, uint32_t n_streams, const jolt_fr_t *tau, size_t n_tau,
                                        const int64_t *a_weights, const int64_t *b_weights, size_t n_nodes, jolt_fr_t *sums_out);
/* spartan_outer / spartan_product remainder (optimized/spartan_outer.rs:236-300,780-850; spartan_product.rs:321-437): Az / Bz at the uni-skip challenge
 * (field weights [stream][1 + n_cols]), the split-eq product member over them, n_tau rounds of degree 3.  output_claims: the n_cols claimed inputs at the cycle point. */
int32_t jolt_stage_spartan_remainder_create(jolt_ctx *ctx, const jolt_ints *const *cols, size_t n_cols, uint32_t n_streams, const jolt_fr_t *a_weights,
                                            const jolt_fr_t *b_weights, const jolt_fr_t *tau, size_t n_tau, const jolt_fr_t *scale, jolt_stage_op **out);
/* ram_read_write (optimized/ram_read_write.rs:58-330): RamAccessColumns (u64 jolt_ints of T entries), RamInc (i64, T), the initial memory (u64, K); log T cycle rounds
 * (cubic, gruen_poly_deg_3) then log K address rounds (quadratic).  output_claims: {ra, val, inc, bound cycle-eq factor}. */
int32_t jolt_stage_ram_read_write_create(jolt_ctx *ctx, const jolt_ints *addresses, const jolt_ints *pre_values, const jolt_ints *post_values, const jolt_ints *inc,
                                         const jolt_ints *val_init, const jolt_fr_t *tau_low, const jolt_fr_t *gamma, jolt_stage_op **out);
/* registers_read_write (optimized/registers_read_write/mod.rs:79-402): `regs` = the hot-index columns rs1, rs2, rd (K = 128), the value columns, RdInc (i128).
 * output_claims: {registers_val, rd_wa, gamma rs1_ra + gamma^2 rs2_ra, rd_inc, bound cycle-eq factor, rs1_ra, rs2_ra}. */
int32_t jolt_stage_registers_read_write_create(jolt_ctx *ctx, const jolt_onehot *regs, const jolt_ints *rs1_val, const jolt_ints *rs2_val, const jolt_ints *rd_pre,
                                               const jolt_ints *rd_post, const jolt_ints *inc, const jolt_fr_t *r_cycle, const jolt_fr_t *gamma, jolt_stage_op **out);
/* booleanity_address (optimized/booleanity.rs:152-427): the pushforward masses of all RA columns against eq(reference_cycle, .) (T-scale, here), then log K rounds
 * of degree 3 over K-entry tables.  Input claim zero.  output_claims: {the phase's intermediate claim}. */
int32_t jolt_stage_booleanity_address_create(jolt_ctx *ctx, const jolt_onehot *cols, const jolt_fr_t *reference_cycle, size_t n_cycle, const jolt_fr_t *reference_address,
                                             const jolt_fr_t *gamma, jolt_stage_op **out);
/* booleanity_cycle (optimized/booleanity.rs:436-690, stage 6b): eq(reference_cycle, j) eq(r_address, reference_address) sum_i (H_i(j)^2 - gamma^i H_i(j)) over the lazily bound
 * RA columns, H_i(j) = gamma^i eq(r_address, hot_i(j)); r_address = the address phase's bound point (log K coordinates, most significant first).  log T rounds of degree 3
 * (gruen_poly_deg_3 of the member's two sums); the input claim is the address phase's intermediate claim (the caller's; no input_claim helper); output claims: the n_polys bound columns unscaled, ra_i(r_address, r_cycle). */
int32_t jolt_stage_booleanity_cycle_create(jolt_ctx *ctx, const jolt_onehot *cols, const jolt_fr_t *r_address, const jolt_fr_t *reference_address, const jolt_fr_t *reference_cycle,
                                           size_t n_cycle, const jolt_fr_t *gamma, jolt_stage_op **out);
/* hamming_weight_claim_reduction (optimized/hamming_weight_claim_reduction.rs:83-300): virtualization_points = n_polys x log K.  output_claims: the bound G_i. */
int32_t jolt_stage_hamming_weight_create(jolt_ctx *ctx, const jolt_onehot *cols, const jolt_fr_t *r_cycle, size_t n_cycle, const jolt_fr_t *r_address,
                                         const jolt_fr_t *virtualization_points, const jolt_fr_t *gamma, jolt_stage_op **out);
/* instruction_read_raf (optimized/instruction_read_raf.rs:736-1456), ONE kernel of 128 + log T rounds: 16 address phases (condensation + scans on the device at each
 * phase's first round, 8 quadratic rounds over 256-entry polynomials on the host) and the cycle rounds over combined * prod ra_i (degree ra_count + 2).
 * claim_columns: the packed flag facts as four K = 16 hot-index columns (tables 0..15, 16..31, 32..41; RAF rows on 0).
 * output_claims: {lookup_table_flags of the present tables, instruction_raf_flag, the ra_count bound ra_i}. */
int32_t jolt_stage_instruction_read_raf_create(jolt_ctx *ctx, jolt_read_raf *rows, const jolt_onehot *claim_columns, const jolt_fr_t *r_reduction, size_t n_vars,
                                               const jolt_fr_t *gamma, const uint8_t *table_present /* 42 */, uint32_t ra_count, jolt_stage_op **out);
/* bytecode_read_raf_address (optimized/bytecode_read_raf.rs:152-437): the five stage pushforwards onto the bytecode domain in one walk over the PC index, log K
 * rounds of degree 2 over 13 K-sized tables.  output_claims: {the 13 bound tables (F_0..4, V_0..4, Int, entry_trace, entry_expected), the intermediate claim}.
 * bytecode_read_raf_cycle (:440-690) is prepared from the finished address operator (its parked eq tables are consumed): log T rounds of C * prod ra_i;
 * output_claims: the bound ra_i. */
int32_t jolt_stage_bytecode_read_raf_address_create(jolt_ctx *ctx, const jolt_key_index *pc_index, const jolt_fr_t *stage_points /* 5 x n_vars */, size_t n_vars,
                                                    const jolt_fr_t *stage_values /* 5 x K */, const jolt_fr_t *gamma, uint64_t first_pc, uint64_t entry_index,
                                                    jolt_stage_op **out);
int32_t jolt_stage_bytecode_read_raf_cycle_create(jolt_ctx *ctx, jolt_stage_op *address, const jolt_onehot *pc_chunks, uint32_t chunk_bits, jolt_stage_op **out);
/* ram_raf_evaluation (optimized/ram_raf_evaluation.rs:17-62): log K rounds of ra_folded * unmap.  output_claims: {ra_folded, unmap} bound. */
int32_t jolt_stage_ram_raf_evaluation_create(jolt_ctx *ctx, const jolt_key_index *ram_index, const jolt_fr_t *tau_low, size_t n_vars, uint64_t lowest_address,
                                             jolt_stage_op **out);
/* ram_output_check (optimized/ram_output_check.rs:50-215): val_final from the access columns (T-scale, here), log K rounds of degree 3.
 * output_claims: {val_final at the bound address point}. */
int32_t jolt_stage_ram_output_check_create(jolt_ctx *ctx, const jolt_key_index *ram_index, const jolt_ints *post_values, const uint64_t *val_init /* K */,
                                           const uint64_t *val_io /* K */, uint64_t io_lo, uint64_t io_len, const jolt_fr_t *r_address, jolt_stage_op **out);
/* Test hook (CPU suite; no device, no context): the reference tier's dense member (NaiveSumcheckProver, crates/jolt-kernels/src/reference/naive.rs:53-377, LowToHigh) over HOST
 * tables as a stage operator, so that the contract and the two drivers below run without a GPU against the oracle's prove_batch.  Tables are copied. */
int32_t jolt_stage_host_expr_create(const jolt_fr_t *const *tables, size_t len, const jolt_member_desc *desc, jolt_stage_op **out);
/* Drivers for the tests and the bench (a Rust host calls the contract above from its own prove_batch; `ctx` may be NULL for host-only operators): prove_batch (prover.rs:193-362) over operators with the
 * library's test transcript; and one operator driven alone -- per round the message, every coefficient absorbed, Transcript::challenge -- with
 * coeffs_out = rounds x stride (tails zeroed), n_coeffs_out[r] = the coefficients of round r, claim in = input claim / out = the final claim. */
int32_t jolt_host_prove_batch_ops(jolt_ctx *ctx, jolt_stage_op *const *ops, size_t n_ops, const jolt_fr_t *input_claims, const jolt_fr_t *coefficients,
                                  const size_t *offsets, size_t max_num_vars, size_t max_degree, uint64_t transcript_label, int32_t challenge_mode,
                                  jolt_fr_t *out_polys, jolt_fr_t *out_challenges, jolt_fr_t *out_member_claims, jolt_fr_t *out_final_claim);
int32_t jolt_host_stage_op_prove_alone(jolt_stage_op *op, jolt_host_transcript *transcript, jolt_fr_t *claim, jolt_fr_t *coeffs_out, size_t stride,
                                       uint32_t *n_coeffs_out, jolt_fr_t *challenges_out);

#ifdef __cplusplus
}
#endif
#endif /* JOLT_HIP_H */
