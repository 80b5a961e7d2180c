// jolt_amd/csrc/small_round.hip.h -- round 0 of a dense member straight off unpromoted 64-bit witness columns, and the first bind into field tables.
//
// The optimized tier of the reference keeps witness columns as compact scalars (Polynomial<T>, T = u8 .. i128) through round 0 -- products of a field element
// with a machine integer go through mul_u64 / FrSmallScalarAccumulator (crates/jolt-field/src/bn254/mont.rs:286-305, 343-427; trait crates/jolt-field/src/
// algebra.rs:362-433) -- and only the FIRST bind produces field elements (Polynomial::bind_to_field, crates/jolt-poly/src/dense.rs:129-142).  Here the same for
// the descriptor-driven member of sumcheck_kernels.hip.h: a table of the member may be a resident u64 column (8 bytes per entry instead of 32, no promotion pass),
//
//   * a linear combination with integer entries is accumulated UNREDUCED, sum_k (c_k R) * v_k in a 13-limb SmallAcc, and reduced ONCE per endpoint (one REDC
//     instead of one Montgomery product per entry: 16 multiply-adds per entry + 64 per REDC against 162 per product);
//   * a product group all of whose factors are single integer columns (flag x value: the shape of instruction_input.rs:1-22) is evaluated EXACTLY in integers at
//     every point (|lo + t (hi - lo)| < 2^67, products < 2^134), multiplied by the group's coefficient inside the accumulator, and all such groups of a pair share
//     one REDC per evaluation point;
//   * the eq weight of an eq-weighted member multiplies the pair's TOTAL once per point;
//   * the first bind writes lo + r (hi - lo) as REDC((1 - r) R^2 * lo + r R^2 * hi): 32 multiply-adds + one REDC per output, 16 bytes read, 32 written.
//
// Work item = pair (row-major): round 0 has >= 2^14 pairs (smaller members are promoted when they are created) and every group of a pair feeds the same REDC.
// The per-pair evaluation is a __host__ __device__ function: tests/test_abi_cpu.py runs it on the host against a big-integer model (jolt_host_small_round_pair)
// -- same descriptor analysis, same accumulator code as the kernel.  Field results are canonical, so the round sums equal the promoted member's bit for bit.
#pragma once
#include "desc.hpp"
#include "small_scalar.hip.h"
#include "sumcheck_kernels.hip.h"

namespace jolt {

// acc += a * m, m = LIMBS <= 5 limbs of 32 bits (a < 2^256: the product is below 2^(256 + 32 LIMBS) <= 2^416)
template <int LIMBS>
JOLT_HD void small_fmadd_n(SmallAcc& acc, const Fr& a, const uint32_t* m) {
#pragma unroll
    for (int i = 0; i < LIMBS; ++i) {
        uint64_t p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (uint64_t)a.l[j] * m[i] + acc.l[i + j];
        uint32_t c = 0;
        acc.l[i] = (uint32_t)p[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) acc.l[i + j] = __builtin_addc((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c, &c);
        acc.l[i + 8] = __builtin_addc(acc.l[i + 8], (uint32_t)(p[7] >> 32), c, &c);
#pragma unroll
        for (int k = i + 9; k < kSmallLimbs; ++k) acc.l[k] = __builtin_addc(acc.l[k], 0u, c, &c);
    }
}
JOLT_HD void small_fmadd_u64(SmallAcc& acc, const Fr& a, uint64_t v) {
    const uint32_t m[2] = {(uint32_t)v, (uint32_t)(v >> 32)};
    small_fmadd_n<2>(acc, a, m);
}

// lo + t (hi - lo) = t hi - (t - 1) lo for u64 endpoints and a point t <= 8, as sign + 96-bit magnitude (< 2^68)
JOLT_HD void int_linear_at(uint64_t lo, uint64_t hi, uint32_t t, uint32_t (&mag)[3], uint32_t& negative) {
    if (t == 0) {
        mag[0] = (uint32_t)lo; mag[1] = (uint32_t)(lo >> 32); mag[2] = 0; negative = 0;
        return;
    }
    // a = t * hi, b = (t - 1) * lo, each below 2^68: three 32-bit limbs
    const uint64_t a0 = (uint64_t)(uint32_t)hi * t, a1 = (hi >> 32) * t + (a0 >> 32);
    const uint64_t b0 = (uint64_t)(uint32_t)lo * (t - 1), b1 = (lo >> 32) * (t - 1) + (b0 >> 32);
    const uint32_t a[3] = {(uint32_t)a0, (uint32_t)a1, (uint32_t)(a1 >> 32)};
    const uint32_t b[3] = {(uint32_t)b0, (uint32_t)b1, (uint32_t)(b1 >> 32)};
    uint32_t br = 0, d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = __builtin_subc(a[i], b[i], br, &br);
    negative = br;  // a < b: the difference is negative, its magnitude the two's complement of d
    const uint32_t mask = 0u - br;
    uint32_t c = br;
#pragma unroll
    for (int i = 0; i < 3; ++i) mag[i] = __builtin_addc(d[i] ^ mask, 0u, c, &c);
}
// 96 x 96 -> 192 bits (the operands here are below 2^68: the top limb of the product is zero)
JOLT_HD void mul_96(const uint32_t (&a)[3], const uint32_t (&b)[3], uint32_t (&p)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) p[i] = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint64_t v = (uint64_t)a[i] * b[j] + p[i + j] + carry;  // <= (2^32 - 1)^2 + 2 (2^32 - 1) = 2^64 - 1
            p[i + j] = (uint32_t)v;
            carry = v >> 32;
        }
        p[i + 3] = (uint32_t)carry;
    }
}

// The member's summand over ONE LowToHigh pair at the NE evaluation points (0, then SKIP1 ? 2, 3, .. : 1, 2, ..): out[s] = sum_g prod_{f in g} factor_f(point_s).
//   ldf(table, lo, hi): the pair of a field table;  ldi(table, lo, hi): the pair of an integer column.
template <int NE, bool SKIP1, class LoadFr, class LoadInt>
JOLT_HD void small_pair_eval(const MemberDesc* __restrict__ d, const SmallDesc* __restrict__ sd, const LoadFr& ldf, const LoadInt& ldi, Fr (&out)[NE]) {
    Fr sum[NE];
    SmallAcc G[NE];
#pragma unroll
    for (int s = 0; s < NE; ++s) { sum[s] = Fr::zero(); G[s] = small_zero(); }
    const uint32_t n_groups = d->n_groups;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t f0 = d->grp_fac_off[g], f1 = d->grp_fac_off[g + 1];
        if (sd->grp_int[g]) {
            uint64_t lo0, hi0, lo1 = 0, hi1 = 0;
            ldi(d->lc_tab[d->fac_lc_off[f0]], lo0, hi0);
            const bool two = f1 - f0 == 2;
            if (two) ldi(d->lc_tab[d->fac_lc_off[f0 + 1]], lo1, hi1);
            const Fr c = sd->grp_coeff_rr[g], nc = neg(c);
#pragma unroll
            for (int s = 0; s < NE; ++s) {
                const uint32_t point = s == 0 ? 0u : (uint32_t)(SKIP1 ? s + 1 : s);
                uint32_t m0[3], n0;
                int_linear_at(lo0, hi0, point, m0, n0);
                if (two) {
                    uint32_t m1[3], n1, p[6];
                    int_linear_at(lo1, hi1, point, m1, n1);
                    mul_96(m0, m1, p);
                    small_fmadd_n<5>(G[s], select((n0 ^ n1) != 0, nc, c), p);
                } else {
                    small_fmadd_n<3>(G[s], select(n0 != 0, nc, c), m0);
                }
            }
            continue;
        }
        Fr prod[NE];
        for (uint32_t f = f0; f < f1; ++f) {
            Fr lo = d->fac_has_const[f] ? d->fac_const[f] : Fr::zero(), hi = lo;
            SmallAcc slo = small_zero(), shi = small_zero();
            bool any_small = false;
            const uint32_t k0 = d->fac_lc_off[f], k1 = d->fac_lc_off[f + 1];
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t ti = d->lc_tab[k];
                if (sd->tab_int[ti]) {
                    uint64_t a, b;
                    ldi(ti, a, b);
                    const Fr c = sd->lc_coeff_rr[k];
                    small_fmadd_u64(slo, c, a);
                    small_fmadd_u64(shi, c, b);
                    any_small = true;
                } else {
                    Fr a, b;
                    ldf(ti, a, b);
                    if (!d->lc_one[k]) {
                        const Fr c = d->lc_coeff[k];
                        a = mul(a, c);
                        b = mul(b, c);
                    }
                    lo = add(lo, a);
                    hi = add(hi, b);
                }
            }
            if (any_small) {
                lo = add(lo, small_redc<FrParams>(slo));
                hi = add(hi, small_redc<FrParams>(shi));
            }
            const Fr step = sub(hi, lo);
            Fr v = lo;
            if (f == f0) {
                prod[0] = v;
                if constexpr (SKIP1) v = add(v, step);
#pragma unroll
                for (int t = 1; t < NE; ++t) { v = add(v, step); prod[t] = v; }
            } else {
                prod[0] = mul(prod[0], v);
                if constexpr (SKIP1) v = add(v, step);
#pragma unroll
                for (int t = 1; t < NE; ++t) { v = add(v, step); prod[t] = mul(prod[t], v); }
            }
        }
        if (f1 == f0) {
#pragma unroll
            for (int t = 0; t < NE; ++t) prod[t] = Fr::one();
        }
#pragma unroll
        for (int t = 0; t < NE; ++t) sum[t] = add(sum[t], prod[t]);
    }
    if (sd->n_int_groups) {
#pragma unroll
        for (int s = 0; s < NE; ++s) sum[s] = add(sum[s], small_redc<FrParams>(G[s]));
    }
#pragma unroll
    for (int s = 0; s < NE; ++s) out[s] = sum[s];
}

struct SmallGroupArgs {
    const SmallDesc* sd[kMaxGroupMembers];
};
// Round 0 of the members of one batch round that still hold integer columns (blockIdx.y = member; LowToHigh, no pending bind).
template <int NE, bool SKIP1>
static __global__ __launch_bounds__(kBlock) void k_round_evals_small(RoundGroupArgs a, SmallGroupArgs s, Fr* __restrict__ partials, RoundDone rd) {
    const int m = blockIdx.y;
    const MemberDesc* __restrict__ d = a.desc[m];
    const SmallDesc* __restrict__ sd = s.sd[m];
    const Fr* const* __restrict__ tabs = a.tabs + a.tab_off[m];
    const size_t half = a.half[m];
    const Fr* __restrict__ e_out = a.e_out[m];
    const Fr* __restrict__ e_in = a.e_in[m];
    const int in_bits = a.in_bits[m];
    Fr acc[NE];
#pragma unroll
    for (int t = 0; t < NE; ++t) acc[t] = Fr::zero();
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t y = (size_t)blockIdx.x * kBlock + threadIdx.x; y < half; y += stride) {
        auto ldf = [&](uint32_t ti, Fr& lo, Fr& hi) {
            lo = ld_fr(tabs[ti] + 2 * y);
            hi = ld_fr(tabs[ti] + 2 * y + 1);
        };
        auto ldi = [&](uint32_t ti, uint64_t& lo, uint64_t& hi) {  // the pair is one aligned 16-byte load
            const uint4 w = reinterpret_cast<const uint4*>(tabs[ti])[y];
            lo = (uint64_t)w.x | ((uint64_t)w.y << 32);
            hi = (uint64_t)w.z | ((uint64_t)w.w << 32);
        };
        Fr tot[NE];
        small_pair_eval<NE, SKIP1>(d, sd, ldf, ldi, tot);
        if (e_out) {
            const Fr e = mul(ld_fr(e_out + (y >> in_bits)), ld_fr(e_in + (y & (((size_t)1 << in_bits) - 1))));
#pragma unroll
            for (int t = 0; t < NE; ++t) tot[t] = mul(tot[t], e);
        }
#pragma unroll
        for (int t = 0; t < NE; ++t) acc[t] = add(acc[t], tot[t]);
    }
    block_reduce_store<NE>(acc, partials + a.part_off[m]);
    finish_member(partials + a.part_off[m], NE, a.ticket[m], a.slot[m], rd);
}

// Polynomial::bind_to_field (crates/jolt-poly/src/dense.rs:129-142) for blockIdx.y-many u64 columns: out[y] = in[2y] + r (in[2y+1] - in[2y]) as a field element,
// a_rr = (1 - r) R^2, b_rr = r R^2 (Montgomery forms of (1 - r) R and r R): REDC(a_rr * lo + b_rr * hi) = ((1 - r) lo + r hi) R.
struct BindIntsBatch {
    const uint64_t* in[kMaxBatchTables];
    Fr* out[kMaxBatchTables];
    size_t half[kMaxBatchTables];
};
static __global__ __launch_bounds__(kBlock) void k_bind_ints_to_field(BindIntsBatch b, Fr a_rr, Fr b_rr) {
    const uint4* __restrict__ in = reinterpret_cast<const uint4*>(b.in[blockIdx.y]);
    Fr* __restrict__ out = b.out[blockIdx.y];
    const size_t half = b.half[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t y = (size_t)blockIdx.x * kBlock + threadIdx.x; y < half; y += stride) {
        const uint4 w = in[y];
        SmallAcc acc = small_zero();
        const uint32_t lo[2] = {w.x, w.y}, hi[2] = {w.z, w.w};
        small_fmadd_n<2>(acc, a_rr, lo);
        small_fmadd_n<2>(acc, b_rr, hi);
        st_fr(out + y, small_redc<FrParams>(acc));
    }
}

}  // namespace jolt
