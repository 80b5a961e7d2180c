"""Synthetic R1CS rows for the Spartan-outer tests (TEST INFRASTRUCTURE): a small random sparse constraint system in the shape of
jolt_r1cs::ConstraintMatrices (rows of (column, coefficient) entries, column 0 = the constant; crates/jolt-r1cs/src/constraint.rs),
random per-(node, stream) row weights standing in for spartan_outer_row_weights, and the per-column weights they fold into
(ConstraintMatrices::weighted_columns + public_column_contributions, reference/spartan_outer.rs:246-256)."""
import numpy as np

from util import rand_fr


def make_system(n_rows, n_inputs, seed):
    rng = np.random.default_rng(seed)
    small = lambda: rand_fr(1, int(rng.integers(1 << 30)))[0]

    def rows():
        out = []
        for _ in range(n_rows):
            k = int(rng.integers(1, min(4, n_inputs + 2)))
            cols = rng.choice(n_inputs + 1, size=k, replace=False)
            out.append([(int(c), small()) for c in cols])
        return out

    return rows(), rows()


def column_weights(rows, row_weights, n_inputs, O):
    """row_weights: (..., n_rows, 4) -> (..., 1 + n_inputs, 4): w_col[c] = sum_r w[r] * A[r][c]"""
    rw = np.asarray(row_weights, dtype=np.uint64)
    lead = rw.shape[:-2]
    flat = rw.reshape(-1, rw.shape[-2], 4)
    out = np.zeros((flat.shape[0], 1 + n_inputs, 4), dtype=np.uint64)
    for i in range(flat.shape[0]):
        for r, row in enumerate(rows):
            for c, a in row:
                out[i, c] = O.fr_add(out[i, c].reshape(1, 4), O.fr_mul(flat[i, r].reshape(1, 4), np.asarray(a).reshape(1, 4)))[0]
    return out.reshape(lead + (1 + n_inputs, 4))
