// jolt_amd/csrc/onehot.hpp -- one-hot (Twist/Shout) selector columns kept as per-cycle hot indices.
//
// The reference never materialises the K x T one-hot grids of the RA polynomials at scale: a selector column
// ra_i(., j) is a point mass at chunk_i(j), so an address-folded column is ONE lookup per cycle into a K-entry scale table
// (crates/jolt-kernels/src/optimized/lazy_ra.rs:1-33, ChunkIndexSource :39-51).  The device twin keeps the hot indices as
// one byte per (polynomial, cycle) -- 0xFF = cold cycle -- instead of 32 bytes of field element.
#pragma once
#include "ctx.hpp"

constexpr uint8_t kOneHotCold = 0xFF;

struct jolt_onehot {
    jolt_ctx* ctx = nullptr;
    uint8_t* idx = nullptr;  // device, [poly][cycle]
    size_t n_polys = 0, cycles = 0;
    uint32_t k = 0;          // scale-table entries (16 or 256; <= 255 because 0xFF marks a cold cycle)
};
