"""Synthetic SpartanProductRow columns (TEST INFRASTRUCTURE; crates/jolt-kernels/src/optimized/spartan_product.rs:62-84) and the column
weights that express the reference's lane formulas (extended_products :113-141, cell() :358-378) in the generic column form of
jolt_r1cs_*_small with n_streams = 1."""
import numpy as np

INPUT_ORDER = ("left_input", "lookup_output", "jump", "right_input", "branch", "next_is_noop")


def make_rows(T, seed):
    rng = np.random.default_rng(seed)
    right = [int(rng.integers(0, 2**63)) * int(rng.integers(0, 2**63)) * (1 if rng.random() < 0.5 else -1) for _ in range(T)]
    right[: min(T, 4)] = [-(2**127), 2**127 - 1, 0, -1][: min(T, 4)]
    rows = {
        "left_input": rng.integers(0, 2**64, size=T, dtype=np.uint64),
        "lookup_output": rng.integers(0, 2**64, size=T, dtype=np.uint64),
        "jump": rng.integers(0, 2, size=T).astype(np.uint8),
        "right_input": np.array([[v & (2**64 - 1), (v >> 64) & (2**64 - 1)] for v in right], dtype=np.uint64).reshape(T, 2),
        "branch": rng.integers(0, 2, size=T).astype(np.uint8),
        "next_is_noop": rng.integers(0, 2, size=T).astype(np.uint8),
    }
    rows["left_input"][: min(T, 2)] = [2**64 - 1, 0][: min(T, 2)]
    rows["_right_python"] = right
    return rows


def device_columns(ctx, rows):
    """the six lanes as jolt_ints columns in INPUT_ORDER (flags widened to u64)"""
    return [ctx.ints(rows["left_input"]), ctx.ints(rows["lookup_output"]), ctx.ints(rows["jump"].astype(np.uint64)), ctx.ints(rows["right_input"], "i128"),
            ctx.ints(rows["branch"].astype(np.uint64)), ctx.ints(rows["next_is_noop"].astype(np.uint64))]


def integer_column_weights(coefficients):
    """coefficients: (nodes, 3) int64 -> A (left) and B (right) weights [node][1][1 + 6]; right lane 2 is 1 - next_is_noop"""
    nodes = coefficients.shape[0]
    a = np.zeros((nodes, 1, 7), dtype=np.int64)
    b = np.zeros((nodes, 1, 7), dtype=np.int64)
    for p in range(nodes):
        c0, c1, c2 = (int(x) for x in coefficients[p])
        a[p, 0, 1:4] = [c0, c1, c2]
        b[p, 0, 0] = c2
        b[p, 0, 4:7] = [c0, c1, -c2]
    return a, b


def field_column_weights(weights, O):
    """weights: (3, 4) field -> A / B weights [1][1 + 6][4] of the remainder's left / right tables"""
    neg = lambda x: O.fr_neg(np.asarray(x).reshape(1, 4))[0]
    a = np.zeros((1, 7, 4), dtype=np.uint64)
    b = np.zeros((1, 7, 4), dtype=np.uint64)
    a[0, 1], a[0, 2], a[0, 3] = weights[0], weights[1], weights[2]
    b[0, 0], b[0, 4], b[0, 5], b[0, 6] = weights[2], weights[0], weights[1], neg(weights[2])
    return a, b
