"""Size-independent checks of a HyperKZG opening whose toxic waste beta is known to the test (TEST INFRASTRUCTURE).

With bases srs[i] = beta^i * G every commitment is commit(p) = p(beta) * G, so without pairings:
  com[i-1] == P_i(beta) * G                                   level commitments (crates/jolt-hyperkzg/src/scheme.rs:141-145)
  w[t]     == ((B(beta) - B(u_t)) / (beta - u_t)) * G          witness commitments (kzg.rs:108-116; B = sum_j q^j P_j, :95-105)
  2 r v2[i+1] == r (1 - x) (v0[i] + v1[i]) + x (v0[i] - v1[i])  the verifier's folding relation (scheme.rs:210-234), v2[ell] = P(point)
  v[t][j] == P_j(u_t)                                          by the oracle's Horner over the oracle's own fold of the downloaded evaluations
The field arithmetic of the checker is the oracle's (tests only).
"""
import numpy as np

import oracle_lib as O


def _m(a, b):
    return O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]


def _a(a, b):
    return O.fr_add(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]


def _s(a, b):
    return O.fr_sub(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]


def same_point(p, q):
    return O.g1_eq(p, q)


def check_opening(ctx, evals_table, point, proof, beta, claimed_eval, max_download_log=22):
    """evals_table: device Table of 2^ell evaluations; proof: dict(com, w, v, challenges) of jolt_host_hyperkzg_open."""
    ell = point.shape[0]
    g = O.g1_generator()
    one = O.to_mont([1])[0]
    r, q = proof["challenges"][0], proof["challenges"][1]
    zero = np.zeros(4, dtype=np.uint64)
    u = [r, _s(zero, r), _m(r, r)]
    v = proof["v"]
    # --- folding relation (anchors every v[.][i] to the claimed evaluation)
    two_r = _a(r, r)
    y_sq = list(v[2]) + [claimed_eval]
    for i in range(ell):
        x = point[ell - 1 - i]
        lhs = _m(two_r, y_sq[i + 1])
        rhs = _a(_m(_m(r, _s(one, x)), _a(v[0][i], v[1][i])), _m(x, _s(v[0][i], v[1][i])))
        assert np.array_equal(lhs, rhs), f"folding relation fails at level {i}"
    # --- P_j(beta) and v[t][j] = P_j(u_t).  When level 0 fits the download bound, EVERY level is the oracle's: level 0 downloaded once, folded by
    # the oracle (scheme.rs:88-114), each device level compared with it entry for entry, Horner by the oracle (kzg.rs:51-59) -- nothing
    # below comes from a device kernel.  Above the bound (callers that cannot afford 32 B x 2^ell on the host) the long levels fall
    # back on the device's blocked Horner, checked against the oracle on the levels that do fit.
    levels = ctx.hyperkzg_fold(evals_table, point)
    p_beta = []
    if len(levels[0]) <= (1 << max_download_log):
        host_levels = O.hyperkzg_fold_polynomials(levels[0].download(), point)
        assert np.array_equal(O.poly_evaluate(host_levels[0], point), claimed_eval), "claimed evaluation differs from the oracle's (dense.rs:340-366)"
        for j, lvl in enumerate(levels):
            assert np.array_equal(lvl.download(), host_levels[j]), f"device fold differs from the oracle at level {j}"
            p_beta.append(O.kzg_eval_univariate(host_levels[j], beta))
            for t in range(3):
                assert np.array_equal(v[t][j], O.kzg_eval_univariate(host_levels[j], u[t])), f"v[{t}][{j}]"
        del host_levels
    else:
        dev_beta = ctx.hyperkzg_eval3(levels, np.stack([beta, beta, beta]))[0]
        for j, lvl in enumerate(levels):
            if len(lvl) <= (1 << max_download_log):
                host = lvl.download()
                pb = O.kzg_eval_univariate(host, beta)
                assert np.array_equal(pb, dev_beta[j]), f"device Horner differs from the oracle at level {j}"
                for t in range(3):
                    assert np.array_equal(v[t][j], O.kzg_eval_univariate(host, u[t])), f"v[{t}][{j}]"
                p_beta.append(pb)
            else:
                p_beta.append(dev_beta[j])
    for lvl in levels:
        lvl.free()
    # --- level commitments
    for i in range(1, ell):
        assert same_point(proof["com"][i - 1], O.g1_scalar_mul(g, p_beta[i])), f"com[{i - 1}]"
    # --- witness commitments: h_t(beta) = (B(beta) - B(u_t)) / (beta - u_t)
    b_beta, qj = zero, one
    b_u = [zero, zero, zero]
    for j in range(ell):
        b_beta = _a(b_beta, _m(qj, p_beta[j]))
        for t in range(3):
            b_u[t] = _a(b_u[t], _m(qj, v[t][j]))
        qj = _m(qj, q)
    for t in range(3):
        den = O.fr_inv(_s(beta, u[t]).reshape(1, 4))[0]
        h_beta = _m(_s(b_beta, b_u[t]), den)
        assert same_point(proof["w"][t], O.g1_scalar_mul(g, h_beta)), f"w[{t}]"
    return p_beta[0]  # P(beta): lets the caller compare the commitment
