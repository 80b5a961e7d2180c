// jolt_amd/csrc/term_map.hip.h -- which terms of a sharded MSM / which coefficients of a sharded polynomial a rank owns, and where they
// sit in its COMPACT arrays.  A rank's compact SRS holds the bases of its terms in index order, and both maps
// are increasing in the slot, so the terms a rank owns of ANY prefix [0, n) are a prefix of its compact arrays: one set of fixed-base
// window tables per rank serves every level of a HyperKZG opening.
//
//   kBlockCyclic  term i belongs to rank (i / block) % world.  With block = the rank's cycle count, a rank's terms are the commitment
//                 grid of ITS cycles (single-GPU layout k * T + j).
//   kSubtree      world = 2^gamma; the gamma bits below the leading one of i name the owner (indices below `world`: owner i, slot 0;
//                 otherwise slot = i with those bits removed).  Additionally closed under LowToHigh folding: slot 2c, 2c + 1 of a
//                 level fold into slot c of the next (c >= 1), so the HyperKZG folds, the RLC and -- with one small exchange each --
//                 the evaluations and the quotient scans run on the compact arrays, 1 / world of the polynomial per rank
//                 (tests/subtree_model.py is the executable specification).
#pragma once
#include <cstddef>
#include <cstdint>

#include "field.hip.h"

namespace jolt {

enum : uint32_t { kTermsAll = 0, kBlockCyclic = 1, kSubtree = 2 };

struct TermMap {
    uint32_t kind = kTermsAll;
    uint32_t gamma = 0;  // kSubtree: log2(world)
    size_t block = 0;    // kBlockCyclic
    size_t rank = 0, world = 1;
};

JOLT_HD int floor_log2_u64(uint64_t x) {  // x >= 1
    int l = 0;
    while (x >> (l + 1)) ++l;
    return l;
}

// compact slot -> global index (increasing in the slot)
JOLT_HD size_t term_global(const TermMap& m, size_t c) {
    if (m.kind == kBlockCyclic) return ((c / m.block) * m.world + m.rank) * m.block + c % m.block;
    if (m.kind == kSubtree) {
        if (c == 0) return m.rank;
        const int lp = floor_log2_u64(c);
        return ((size_t)1 << (lp + m.gamma)) | (m.rank << lp) | (c - ((size_t)1 << lp));
    }
    return c;
}

// global index -> does this rank own it, and in which slot
JOLT_HD bool term_slot(const TermMap& m, size_t i, size_t* c) {
    if (m.kind == kBlockCyclic) {
        const size_t b = i / m.block;
        *c = (b / m.world) * m.block + i % m.block;
        return b % m.world == m.rank;
    }
    if (m.kind == kSubtree) {
        if (i < m.world) { *c = 0; return i == m.rank; }
        const int lp = floor_log2_u64(i) - (int)m.gamma;
        *c = ((size_t)1 << lp) | (i & (((size_t)1 << lp) - 1));
        return ((i >> lp) & (m.world - 1)) == m.rank;
    }
    *c = i;
    return true;
}

// how many of the terms [0, n) the rank owns = the length of its compact prefix
inline size_t term_owned(const TermMap& m, size_t n) {
    if (m.kind == kTermsAll) return n;
    if (m.kind == kBlockCyclic) {
        const size_t full = n / m.block, rem = n % m.block;
        return (full / m.world + (m.rank < full % m.world ? 1 : 0)) * m.block + (full % m.world == m.rank ? rem : 0);
    }
    size_t lo = 0, hi = n + 1;  // first slot whose index is >= n
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (term_global(m, mid) < n) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// the maps of one rank (false: not a valid rank / world / block)
inline bool make_subtree_map(int32_t rank, int32_t world, TermMap* m) {
    if (world < 1 || rank < 0 || rank >= world || (world & (world - 1)) != 0) return false;
    m->kind = kSubtree;
    m->gamma = (uint32_t)floor_log2_u64((uint64_t)world);
    m->block = 0;
    m->rank = (size_t)rank;
    m->world = (size_t)world;
    return true;
}
inline bool make_block_map(size_t block, int32_t rank, int32_t world, TermMap* m) {
    if (world < 1 || rank < 0 || rank >= world || block == 0) return false;
    m->kind = kBlockCyclic;
    m->gamma = 0;
    m->block = block;
    m->rank = (size_t)rank;
    m->world = (size_t)world;
    return true;
}

}  // namespace jolt
