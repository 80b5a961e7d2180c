// jolt_amd/csrc/desc.hpp -- plain descriptor structs shared by the kernels (device) and the member objects (host).
#pragma once
#include "field.hip.h"

namespace jolt {

constexpr int kBlock = 256;
constexpr int kMaxBatchTables = 40;
constexpr int kMaxGroups = 16;
constexpr int kMaxFactors = 64;
constexpr int kMaxLc = 96;

// Device-resident descriptor (read through the scalar cache: every access is wave-uniform).
struct MemberDesc {
    uint32_t n_groups;
    uint32_t n_factors;
    uint32_t n_lc;
    uint32_t grp_fac_off[kMaxGroups + 1];
    uint32_t fac_lc_off[kMaxFactors + 1];
    uint32_t fac_has_const[kMaxFactors];
    uint32_t lc_tab[kMaxLc];
    uint32_t lc_one[kMaxLc];  // coefficient == 1 -> skip the multiply
    uint32_t lc_owner[kMaxLc];  // first entry mentioning its table: in a fused bind+evaluate round it stores the bound values
    Fr fac_const[kMaxFactors];
    Fr lc_coeff[kMaxLc];
};
// What jolt_member_create_lc_small derives from the descriptor of a member some of whose tables are unpromoted u64 columns (small_round.hip.h)
struct SmallDesc {
    uint32_t tab_int[kMaxBatchTables];  // 1: the table is a u64 column (until the first bind)
    uint32_t grp_int[kMaxGroups];       // 1: integer group -- one or two factors, each ONE integer column without a constant
    uint32_t n_int_groups;
    uint32_t pad_[3];
    Fr grp_coeff_rr[kMaxGroups];        // the integer group's coefficient c (the product of its entries' coefficients) as c * R^2: REDC(c R^2 * I) = c I R
    Fr lc_coeff_rr[kMaxLc];             // the same for every LC entry (used where the entry's table is an integer column inside a field-valued factor)
};
// current evaluation buffers of the member's tables (they ping-pong on every LowToHigh bind), passed by value
struct TablePtrs {
    const Fr* p[kMaxBatchTables];
};


}  // namespace jolt
