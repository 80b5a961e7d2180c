// jolt_amd/csrc/onehot.hpp -- one-hot (Twist/Shout) selector columns kept as per-cycle hot indices.
//
// The reference never materialises the K x T one-hot grids of the RA polynomials at scale: a selector column
// ra_i(., j) is a point mass at chunk_i(j), so an address-folded column is ONE lookup per cycle into a K-entry scale table
// (crates/jolt-kernels/src/optimized/lazy_ra.rs:1-33, ChunkIndexSource :39-51).  The device twin keeps the hot indices as
// one byte per (polynomial, cycle) -- 0xFF = cold cycle -- instead of 32 bytes of field element.
#pragma once
#include "ctx.hpp"

constexpr uint8_t kOneHotCold = 0xFF;         // narrow sources: one byte per (polynomial, cycle), k <= 255
constexpr uint16_t kOneHotCold16 = 0xFFFF;    // wide sources: two bytes, k <= 65535 (log_k_chunk = 8 at log T >= 25: K = 256,
                                              // crates/jolt-prover/src/config.rs:175-186)
constexpr uint32_t kColdIdx = 0xFFFFFFFFu;    // what hot_load returns for a cold cycle of either width

struct jolt_onehot {
    jolt_ctx* ctx = nullptr;
    uint8_t* idx = nullptr;  // device, [poly][cycle], 1 << wide bytes per entry
    size_t n_polys = 0, cycles = 0;
    uint32_t k = 0;          // scale-table entries
    uint32_t wide = 0;       // 0: uint8 indices (0xFF = cold), 1: uint16 indices (0xFFFF = cold)
};

#if defined(__HIPCC__)
// entry i of a hot-index array of either width; cold cycles come back as kColdIdx
__device__ __forceinline__ uint32_t hot_load(const uint8_t* __restrict__ base, size_t i, uint32_t wide) {
    if (wide) {
        const uint32_t v = reinterpret_cast<const uint16_t*>(base)[i];
        return v == kOneHotCold16 ? kColdIdx : v;
    }
    const uint32_t v = base[i];
    return v == kOneHotCold ? kColdIdx : v;
}
// start of column `entry_offset / cycles` (offset counted in ENTRIES)
__device__ __forceinline__ const uint8_t* hot_col(const uint8_t* __restrict__ base, size_t entry_offset, uint32_t wide) { return base + (entry_offset << wide); }
#endif
