"""oracle/address_ops.c (the pushforwards of cycle weights onto bytecode / RAM address domains, the final-memory column) and the address-domain drivers of
jolt_amd/stages.py on the CPU oracle.  The reference holds no vectors for these; what it holds is (i) its own membership test of stage_pushforwards against the
naive pushforward over the full eq tables (crates/jolt-kernels/src/optimized/bytecode_read_raf.rs:726-774), re-run here on its inputs, and (ii) the byte-parity of
the optimized kernels with the dense reference-tier members, which the drivers use directly; so the drivers are pinned by the sumcheck round check against claims
taken from the DENSE definitions and by the closed forms of their final claims."""
import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import stages as S
from jolt_amd.workload import rand_fr
from workload_oracle import OracleExtended


def _add(a, b): return O.fr_add(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
def _sub(a, b): return O.fr_sub(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
def _mul(a, b): return O.fr_mul(np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]


ZERO, ONE = np.zeros(4, dtype=np.uint64), O.to_mont([1])[0]


def check_rounds(out, claim):
    """every message sums to the running claim; returns the last claim"""
    for rnd, poly in enumerate(out["polys"]):
        assert np.array_equal(_add(O.univariate_evaluate(poly, ZERO), O.univariate_evaluate(poly, ONE)), claim), rnd
        claim = O.univariate_evaluate(poly, out["challenges"][rnd])
    assert np.array_equal(claim, out["final_claim"])
    return claim


def test_stage_pushforwards_on_the_reference_membership_vector():
    """bytecode_read_raf.rs:735-774: a Boolean low-point coordinate makes stage 0's eq weight exactly zero on half the low domain, so a repeated PC whose first
    visit carried weight zero must not be counted twice"""
    rng = np.random.default_rng(101)
    points = np.stack([rand_fr(4, rng) for _ in range(5)])
    points[0][2] = ONE
    pcs = np.array([2, 0, 2, 1, 3, 3, 0, 2, 1, 1, 1, 1, 0, 3, 2, 0], dtype=np.uint64)
    got = O.stage_pushforwards(points, pcs, 4)
    for s in range(5):
        eq = O.eq_evals(points[s])
        want = np.zeros((4, 4), dtype=np.uint64)
        for j, pc in enumerate(pcs):
            want[int(pc)] = _add(want[int(pc)], eq[j])
        assert np.array_equal(got[s], want), s


@pytest.mark.parametrize("log_t,k", [(1, 2), (5, 8), (7, 1), (8, 300), (9, 4096)])
def test_split_pushforward_equals_fold_cycles(log_t, k):
    rng = np.random.default_rng(log_t * 31 + k)
    points = np.stack([rand_fr(log_t, rng) for _ in range(3)])
    pcs = rng.integers(0, k, size=1 << log_t).astype(np.uint64)
    pcs[rng.random(1 << log_t) < 0.5] = pcs[0]  # one hot PC
    got = O.stage_pushforwards(points, pcs, k)
    for s in range(3):
        assert np.array_equal(got[s], O.fold_cycles(pcs, k, O.eq_evals(points[s])))
    with pytest.raises(ValueError):
        O.stage_pushforwards(points, np.where(np.arange(1 << log_t) == 1, k, pcs).astype(np.uint64), k)


def test_fold_cycles_skips_cold_cycles_and_last_value_is_the_final_memory():
    rng = np.random.default_rng(5)
    tr = S.consistent_ram_trace(4, 8, rng)
    w = rand_fr(1 << 8, rng)
    got = O.fold_cycles(tr["addresses"], 16, w)
    want = np.zeros((16, 4), dtype=np.uint64)
    mem = tr["val_init"].copy()
    for j, a in enumerate(tr["addresses"]):
        if a != S.NO_ACCESS:
            want[int(a)] = _add(want[int(a)], w[j])
            mem[int(a)] = tr["post"][j]
    assert np.array_equal(got, want)
    assert np.array_equal(O.last_value(tr["addresses"], tr["post"], 16, O.fr_from_u64(tr["val_init"])), O.fr_from_u64(mem))
    assert np.array_equal(_add(got.sum(axis=0) * 0, ZERO), ZERO)  # (shape check of the accumulator)


@pytest.mark.parametrize("n_vars,kw", [(5, dict(log_k=3, log_kb=5)), (6, dict(log_k=6, log_kb=4)), (8, dict(log_k=5)), (3, dict(log_k=1, log_kb=2))])
def test_address_domain_drivers_against_the_dense_definitions(n_vars, kw):
    ext = OracleExtended(n_vars, seed=17, n_tables=4, **kw)
    d = ext.d
    out = ext.address_domain(label=40)
    # ---- bytecode, address phase: claim = sum_j sum_s g^s eq_s(j) (Val_s(pc_j) + raf_s pc_j) + g^7 [pc_0 = entry]  (the pushforward summed against the address tables)
    bc = d["bytecode"]
    K = 1 << bc["log_k"]
    gp = [ONE]
    for _ in range(7):
        gp.append(_mul(gp[-1], bc["gamma"]))
    claim = _mul(gp[7], ONE if int(bc["push_pc"][0]) == bc["entry_index"] else ZERO)
    pcs = bc["push_pc"].astype(np.int64)
    pc_f = O.fr_from_u64(bc["push_pc"])
    for s in range(5):
        eq = O.eq_evals(bc["stage_points"][s]) if n_vars else O.to_mont([1])
        val = bc["stage_values"][s][pcs]
        raf = gp[5] if s == 0 else (gp[4] if s == 2 else None)
        if raf is not None:
            val = O.fr_add(val, O.fr_mul(np.repeat(raf.reshape(1, 4), pc_f.shape[0], axis=0), pc_f))
        term = O.Member.expr([eq, val], [(gp[s], [0, 1])], 2)
        claim = _add(claim, term.input_claim())
        term.close()
    b = out["bytecode_read_raf"]
    assert np.array_equal(b["claim_address"], claim)
    assert np.array_equal(check_rounds(b["address"], claim), b["intermediate"])
    for s in range(5):
        assert np.array_equal(b["val_stages"][s], O.poly_evaluate(bc["stage_values"][s], b["r_address"]))
    # ---- bytecode, cycle phase: the last claim is C(r_cycle) * prod_i ra_i(r_cycle); prod_i ra_i(j) = eq(r_address, pc_j) on mapped rows, 0 on cold ones
    last = check_rounds(b["cycle"], b["claim_cycle"])
    r_cycle = b["cycle"]["challenges"][::-1]
    eq_adr = O.eq_evals(b["r_address"])
    prod = np.where(bc["mapped"][:, None], eq_adr[pcs], 0).astype(np.uint64)
    chunks = S.committed_address_chunks(b["r_address"], bc["chunk_bits"])
    ra_cols = [O.onehot_values(O.eq_evals(c), 1, 1 << bc["chunk_bits"], bc["chunk_cols"][i], 1 << n_vars) for i, c in enumerate(chunks)]
    acc = ra_cols[0]
    for col in ra_cols[1:]:
        acc = O.fr_mul(acc, col)
    assert np.array_equal(acc, prod)  # the chunk decomposition reassembles eq(r_address, pc)
    for i, col in enumerate(ra_cols):
        assert np.array_equal(b["ra_claims"][i], O.poly_evaluate(col, r_cycle) if n_vars else col[0])
    # ---- RAM RAF evaluation: claim = sum_j eq(tau_low, j) [access_j] (8 address_j + lowest); last claim = folded(r) unmap(r)
    ram, raf = d["ram"], d["ram_raf"]
    acc_rows = ram["addresses"] != S.NO_ACCESS
    unmapped = np.where(acc_rows, np.uint64(8) * np.where(acc_rows, ram["addresses"], 0) + raf["lowest_address"], 0).astype(np.uint64)
    want = O.Member.expr([O.eq_evals(raf["tau_low"]), O.fr_from_u64(unmapped)], [(ONE, [0, 1])], 2)
    r = out["ram_raf_evaluation"]
    assert np.array_equal(r["claim"], want.input_claim())
    want.close()
    last = check_rounds(r, r["claim"])
    point = r["challenges"][::-1]
    Kr = 1 << ram["log_k"]
    unmap = O.fr_from_u64(np.uint64(8) * np.arange(Kr, dtype=np.uint64) + raf["lowest_address"])
    assert np.array_equal(last, _mul(r["ra_claim"], O.poly_evaluate(unmap, point)))
    assert np.array_equal(r["ra_claim"], O.poly_evaluate(O.fold_cycles(ram["addresses"], Kr, O.eq_evals(raf["tau_low"])), point))
    # ---- RAM output check: claim = sum over the IO words of eq(point, k) (val_final(k) - val_io(k)); last claim = eq(point, r) mask(r) (val_final(r) - val_io(r))
    io = d["ram_output"]
    mem = ram["val_init"].copy()
    for j, a in enumerate(ram["addresses"]):
        if a != S.NO_ACCESS:
            mem[int(a)] = ram["post"][j]
    eq_k = O.eq_evals(io["point"])
    want = ZERO
    for k in range(io["io_lo"], io["io_lo"] + io["io_len"]):
        want = _add(want, _mul(eq_k[k], _sub(O.fr_from_u64([mem[k]])[0], O.fr_from_u64([io["val_io"][k]])[0])))
    oc = out["ram_output_check"]
    assert np.array_equal(oc["claim"], want)
    last = check_rounds(oc, oc["claim"])
    point = oc["challenges"][::-1]
    mask = np.zeros(Kr, dtype=np.uint64)
    mask[io["io_lo"]:io["io_lo"] + io["io_len"]] = 1
    assert np.array_equal(oc["val_final_claim"], O.poly_evaluate(O.fr_from_u64(mem), point))
    closed = _mul(O.eq_mle(io["point"], point), _mul(O.poly_evaluate(O.fr_from_u64(mask), point), _sub(oc["val_final_claim"], O.poly_evaluate(O.fr_from_u64(io["val_io"]), point))))
    assert np.array_equal(last, closed)


@pytest.mark.parametrize("n_vars", [3, 5, 8])
def test_hamming_weight_reduction_against_its_definition(n_vars):
    """stage 7 (optimized/hamming_weight_claim_reduction.rs): claim = sum_i sum_k G_i(k) W_i(k) with G_i the dense pushforward of the column; every message sums to the
    running claim; the last claim is sum_i G_i(r) W_i(r) with W_i(r) = g^(3i) + g^(3i+1) eq(r_address, r) + g^(3i+2) eq(virt_i, r)"""
    ext = OracleExtended(n_vars, seed=23, n_tables=4)
    bo, hw = ext.d["booleanity"], ext.d["hamming"]
    out = ext.hamming_weight(label=60)
    K, n = 1 << bo["log_k"], bo["cols"].shape[0]
    eq = O.eq_evals(hw["r_cycle"])
    gp = [ONE]
    for _ in range(3 * n):
        gp.append(_mul(gp[-1], hw["gamma"]))
    eq_bool = O.eq_evals(hw["r_address"])
    claim = ZERO
    for i in range(n):
        g = np.zeros((K, 4), dtype=np.uint64)
        for j, k in enumerate(bo["cols"][i]):
            if k != 0xFF:
                g[int(k)] = _add(g[int(k)], eq[j])
        assert np.array_equal(out["masses"][i], g)
        eq_virt = O.eq_evals(hw["virtualization_points"][i])
        for k in range(K):
            claim = _add(claim, _mul(g[k], _add(gp[3 * i], _add(_mul(gp[3 * i + 1], eq_bool[k]), _mul(gp[3 * i + 2], eq_virt[k])))))
    assert np.array_equal(out["claim"], claim)
    last = check_rounds(out, claim)
    point = out["challenges"][::-1]
    closed = ZERO
    for i in range(n):
        w_r = _add(gp[3 * i], _add(_mul(gp[3 * i + 1], O.eq_mle(hw["r_address"], point)), _mul(gp[3 * i + 2], O.eq_mle(hw["virtualization_points"][i], point))))
        assert np.array_equal(out["g_claims"][i], O.poly_evaluate(out["masses"][i], point))
        closed = _add(closed, _mul(out["g_claims"][i], w_r))
    assert np.array_equal(last, closed)


def test_host_hamming_helpers_of_the_library_equal_the_oracle():
    """jolt_host_hamming_weights / jolt_host_pair_tables_* (host code of libjolt_hip.so: no GPU needed) against oracle/onehot.c, round for round"""
    from jolt_amd import ffi
    rng = np.random.default_rng(9)
    for n, log_k in [(5, 2), (36, 4), (3, 1)]:
        K = 1 << log_k
        masses = rand_fr(n * K, rng).reshape(n, K, 4)
        gamma, r_address, virt = rand_fr(1, rng)[0], rand_fr(log_k, rng), rand_fr(n * log_k, rng).reshape(n, log_k, 4)
        host, orc = ffi.HostHammingWeight(masses, gamma, r_address, virt), O.HammingWeight(masses, gamma, r_address, virt)
        assert np.array_equal(host.w, orc.w)
        for _ in range(log_k):
            assert np.array_equal(host.round(), orc.round())
            r = rand_fr(1, rng)[0]
            host.bind(r)
            orc.bind(r)
        assert np.array_equal(host.round()[2], orc.round()[2])
        assert np.array_equal(host.output_claims(), orc.output_claims())
