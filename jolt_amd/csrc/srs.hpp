// jolt_amd/csrc/srs.hpp -- the device-resident G1 bases behind a jolt_srs handle (affine, converted once at upload).
#pragma once
#include "ctx.hpp"
#include "g1.cuh"

struct jolt_srs {
    jolt_ctx* ctx = nullptr;
    jolt::G1Affine* pts = nullptr;
    size_t n = 0;
};
