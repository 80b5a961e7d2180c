// jolt_amd/csrc/srs.hpp -- the device-resident G1 bases behind a jolt_srs handle (affine, converted once at upload).
#pragma once
#include "ctx.hpp"
#include "g1.hip.h"

struct jolt_srs {
    jolt_ctx* ctx = nullptr;
    jolt::G1Affine* pts = nullptr;
    size_t n = 0;
    // Fixed-base tables (jolt_srs_precompute_windows, msm_fixed.hip): pre[w * n + i] = 2^(pre_c * w) * pts[i] for w < pre_W, so that
    // every c-bit window of a scalar addresses the SAME bucket set (one bucket reduction per MSM instead of one per window, which
    // is what makes 24-bit windows affordable).  pre_W * n * 64 bytes: the table is sized for 288 GB of HBM, not for a PCIe card.
    jolt::G1Affine* pre = nullptr;
    int pre_c = 0, pre_W = 0;
    bool pre_lform = false;  // the tables hold L-form coordinates (fq_limb.hip.h): x * 2^261 mod p
    uint32_t pre_B = 0;     // bucket count = the largest digit magnitude (k_fx_digits)
    size_t pre_stride = 0;  // points per window table (= n of the SRS the tables were built for; a range view keeps the parent's)
    size_t pre_min_n = 0;  // MSMs shorter than this keep the per-window bucket method
    // A second table set over the first 2^23 bases with 20-bit windows (13 tables, 6.5 GiB), next to a main set sized for >= 2^25 points: the reduction over the main set's
    // 6.3 M buckets costs 3.5 ms whatever the MSM's length -- more than half of a 2^21-term MSM -- so MSMs of 2^19 .. 2^23 terms (the level commitments of an opening below
    // 2^24, the commitments of the 64-bit witness columns) take 2^19 buckets and 13 windows instead of 6.3 M and 11.  Owned; range views with an offset do not see it.
    jolt_srs* mid_tables = nullptr;
};

// bases [offset, n) of `parent` as an SRS of their own (no ownership): term-range MSMs of a sharded opening
static inline jolt_srs jolt_srs_range_view(const jolt_srs& parent, size_t offset) {
    jolt_srs v = parent;
    v.pts = parent.pts + offset;
    v.n = parent.n - offset;
    if (parent.pre) v.pre = parent.pre + offset;
    if (offset) v.mid_tables = nullptr;  // the mid set covers the bases from 0
    return v;
}
