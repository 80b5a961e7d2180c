// jolt_amd/csrc/ints.hpp -- device-resident machine integers behind a jolt_ints handle (witness columns before promotion:
// u64 / i64 / i128 little-endian, the compact scalars of Polynomial<T>, crates/jolt-poly/src/dense.rs:129-142).
#pragma once
#include "ctx.hpp"

struct jolt_ints {
    jolt_ctx* ctx = nullptr;
    void* data = nullptr;  // device
    size_t count = 0;
    int32_t kind = 0;      // JOLT_INT_*
};
