// jolt_amd/csrc/grid_hint.hpp -- the opening hint of the commitment grid (pcs.hip builds it, hyperkzg.hip consumes it).
//
// CommitmentScheme::commit returns (Commitment, OpeningHint) (crates/jolt-openings/src/schemes.rs:60-72): whatever the committer can precompute for the opening
// from the polynomial and the setup alone.  For the one-hot columns of the commitment grid that is, per fold depth s = 1 .. levels, the residue-class sums
//   S_p^(s, c) = sum over the cycles j = c mod 2^s of srs[(hot_p(j) * T + j) >> s],
// with which the first `levels` level commitments of a HyperKZG opening of ANY random linear combination of the columns follow by linearity (section 3.7b of
// docs/kernels.md) -- sums of bases at the commit leg's rate instead of MSMs over 2^(ell - s) full-width scalars, and nothing in them depends on a challenge: they are
// enqueued at commit time on a low-priority stream at one wavefront per SIMD and run under the latency-bound legs between the commitment and the opening.
#pragma once
#include "ctx.hpp"
#include "g1.hip.h"

struct jolt_grid_hint {
    jolt_ctx* ctx = nullptr;
    uint32_t levels = 0, k = 0;
    size_t n_cols = 0, cycles = 0;
    jolt::G1Jac* sums = nullptr;     // device: level s (1-based) at sums + level_offset[s - 1], laid out [class c < 2^s][column]
    size_t level_offset[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // in points; level_offset[levels] = the total
    jolt::G1Jac* partial = nullptr;  // device scratch of the sums' first stage
    hipStream_t stream = nullptr;    // where the sums were enqueued (the context's hint stream, or its main stream)
    hipEvent_t ready = nullptr;      // recorded behind the last sum
};

// d_out[s - 1] = sum_c sum_p (onehot_scalars[p] * w_c^(s)) S_p^(s, c), w_c^(s) = prod_{b < s} (bit b of c ? xs[b] : 1 - xs[b]) -- the one-hot part of the level-s commitment
// of the opening whose fold b uses xs[b]; enqueued on the hint's stream (which first joins the context's main stream), d_out from the context's pool (levels points);
// the caller synchronises with the hint's stream before reading d_out and frees d_out and *d_temp (the uploaded scalars) afterwards.
int32_t jolt_internal_grid_hint_combine(jolt_ctx* ctx, const jolt_grid_hint* hint, uint32_t levels, const jolt::Fr* onehot_scalars, const jolt::Fr* xs, jolt::G1Jac** d_out,
                                        void** d_temp);
