"""The stage 1 / 2 / 4 / 5 / 6a / 6b operators of jolt_amd/stages.py on the device against the same drivers on the CPU oracle (tests/workload_oracle.py
OracleExtended): uni-skip sums, challenges, every round polynomial, claimed inputs, every phase's scans, final values -- message for message,
the reference's optimized-vs-reference lock step (crates/jolt-kernels/src/optimized/parity.rs:79-118) over whole stage drivers."""
import numpy as np
import pytest

from jolt_amd import ffi
from jolt_amd.stages import DeviceExtended
from workload_oracle import OracleExtended

pytestmark = pytest.mark.gpu


def same(a, b, path=""):
    if isinstance(a, dict):
        for k in a:
            same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, f"{path}[{i}]")
    else:
        assert np.array_equal(np.asarray(a), np.asarray(b)), path


def run(n_vars, seed, **kw):
    if n_vars >= 13:  # the oracle's T-scale sweeps on one thread per physical core of the GPU box's host (the OpenMP default, every hardware thread, is slower there)
        import os
        import oracle_lib as O
        O.baseline_set_threads(min(128, os.cpu_count() or 1))
    ctx = ffi.Context(0)
    dev = DeviceExtended(ctx, n_vars, seed=seed, **kw)
    got = dev.prove(label=40)
    again = dev.prove(label=40)  # a second proof over the resident inputs: same bytes
    want = OracleExtended(n_vars, seed=seed, **kw).prove(label=40)
    address_domain = {"bytecode_read_raf", "ram_raf_evaluation", "ram_output_check", "hamming_weight", "booleanity_cycle"}  # their claims travel inside the driver's output
    for name in got:
        if name not in address_domain and name != "booleanity_address":  # (booleanity's input claim is zero by construction)
            assert np.array_equal(dev.claims[{"spartan_outer": "outer", "spartan_product": "product", "ram_read_write": "ram", "registers_read_write": "registers",
                                              "instruction_read_raf": "lookup"}[name]], want[name]["claim"]), name
        same(got[name], {k: v for k, v in want[name].items() if k != "claim" or name in address_domain}, name)
        same(again[name], got[name], name + " (second proof)")
    dev.close()
    ctx.close()


@pytest.mark.parametrize("n_vars,kw", [(6, dict(n_tables=6, log_k=4)), (9, dict(n_tables=12)), (12, dict(log_k=14)), (3, dict(n_tables=3, n_outer=5, n_nodes=3, log_k=2)),
                                       (10, dict(n_tables=5, log_k=6, log_kb=5)), (17, dict(n_tables=8, log_k=17, log_kb=16))])
def test_extended_stages_match_oracle(n_vars, kw):
    run(n_vars, 21 + n_vars, **kw)


def test_extended_stages_match_oracle_at_trace_scale():
    """T = 2^20 cycles (K = 2^16 RAM words, 42 lookup tables, 35 R1CS inputs): the sizes at which the scans run many workgroups per bin, the
    sparse matrix merges hundreds of columns per group and the uni-skip kernel streams the integer columns -- transcript for transcript.
    Instruction read-RAF's 128 address rounds at this size: the 17 rounds of phases 0 and 15 and round 62 are the oracle's FROM THE DEFINITION and must equal the
    product's; the other 111 are sumcheck-ends-verified (OracleExtended.instruction_read_raf), not lock step.  T = 2^22: tests/test_gpu_extended_t22.py."""
    import oracle_lib as O
    import os
    O.baseline_set_threads(min(128, os.cpu_count() or 1))
    run(20, 2026)


def test_instruction_read_raf_every_address_round_from_the_definition_at_2_16():
    """Lock step of ALL 128 address rounds at a mid size (round-4 review, item 9).  Above T = 2^12 the oracle twin normally takes the address-round polynomials from the
    product's host state machine (fed with oracle scan sums) and recomputes only a sample from the definition; here, at T = 2^16 (16 x 1024-row work items per bin, the
    sizes at which the device scans run many workgroups per bin), every one of the 128 polynomials is the oracle's FROM THE DEFINITION -- evaluate_mle of each row's table
    at the mixed point, no prefix / suffix machinery, no product code (oracle/lookup_tables.c, OpenMP) -- and the device path (16 phase scans + condensations on the GPU,
    128 rounds on the library's host side) must produce the same 128 messages, challenges, end values, cycle rounds and output claims."""
    import os
    import oracle_lib as O
    from jolt_amd.stages import build_extended

    class EveryRoundDirect(OracleExtended):
        DIRECT_ADDRESS_ROUNDS_MAX_LOG_T = 16

    O.baseline_set_threads(min(128, os.cpu_count() or 1))
    n_vars, label = 16, 40
    d = build_extended(n_vars, 916)
    ctx = ffi.Context(0)
    dev = DeviceExtended(ctx, n_vars, description=d)
    orc = EveryRoundDirect(n_vars, description=d)
    got = dev.instruction_read_raf(label + 400)
    want = orc.instruction_read_raf(label + 400)
    assert orc.direct_checked == list(range(128))
    assert np.array_equal(dev.claims["lookup"], want["claim"])
    same(got, {k: v for k, v in want.items() if k != "claim"}, "instruction_read_raf")
    dev.close()
    ctx.close()


def test_a_stage_batch_over_operators_is_the_batch_over_their_members():
    """jolt_host_prove_batch_ops with MORE than one operator, as a stage driver batches the members of a stage (prover.rs:193-362): RAM RAF evaluation (degree 2) and the RAM
    output check (degree 3, split-eq) -- both log K rounds -- and booleanity's address phase (log K_chunk rounds, a later window of the batch: front-loaded inactive rounds)
    under one transcript and random batching coefficients.  The first two are dense members underneath, so the batch over the OPERATORS must equal jolt_host_prove_batch over
    the same members built by hand (the round-5 path); prove_batch checks s(0) + s(1) against the running claim every round for all three."""
    from jolt_amd import stages as S
    from util import rand_fr
    n_vars = 10
    ctx = ffi.Context(0)
    dev = DeviceExtended(ctx, n_vars, seed=31, n_tables=5, log_k=6, log_kb=5)
    d = dev.d
    ram, raf, io, bo = d["ram"], d["ram_raf"], d["ram_output"], d["booleanity"]
    index = ctx.key_index(dev.ram_cols[0], 1 << ram["log_k"])
    coeffs = rand_fr(3, 77)

    def operators():
        return [ctx.stage_ram_raf_evaluation(index, raf["tau_low"], raf["lowest_address"]),
                ctx.stage_ram_output_check(index, dev.ram_cols[2], ram["val_init"], io["val_io"], io["io_lo"], io["io_len"], io["point"]),
                ctx.stage_booleanity_address(dev.bool_cols, bo["reference_cycle"], bo["reference_address"], bo["gamma"])]
    ops = operators()
    claims = [ops[0].input_claim(), ops[1].input_claim(), np.zeros(4, dtype=np.uint64)]
    rounds, offsets = ram["log_k"], [0, 0, ram["log_k"] - bo["log_k"]]
    got = ctx.prove_batch_ops(ops, claims, coeffs, offsets, rounds, 3, label=5)
    # the same two dense members by hand (stages.ram_raf_evaluation / ram_output_check, the round-5 drivers), batched by jolt_host_prove_batch; booleanity's operator alone
    # in a second batch of its own cannot share the transcript, so only the two-member prefix is compared message for message: batch them without the third operator
    two = operators()[:2]
    ref_ops = ctx.prove_batch_ops(two, claims[:2], coeffs[:2], [0, 0], rounds, 3, label=6)
    dops = S.DeviceOps(ctx, ffi, {"ram": index, "ram_post": dev.ram_cols[2]}, dev.pc_chunks)
    eq = dops.eq(raf["tau_low"])
    folded = dops.pushforward("ram", [eq])[0]
    eq.free()
    m_raf = dops.member_expr([folded, dops.u64_table(np.uint64(8) * np.arange(1 << ram["log_k"], dtype=np.uint64) + raf["lowest_address"])], [(dops.one, [0, 1])], 2)
    init = dops.u64_table(ram["val_init"])
    val_final = dops.last_value("ram", init)
    val_io = dops.u64_table(io["val_io"])
    diff = dops.rlc([val_final, val_io], [dops.one, dops.sub(np.zeros(4, dtype=np.uint64), dops.one)])
    mask = np.zeros(1 << ram["log_k"], dtype=np.uint64)
    mask[io["io_lo"]:io["io_lo"] + io["io_len"]] = 1
    m_oc = dops.member_gruen_product(dops.u64_table(mask), diff, io["point"])
    want = ctx.prove_batch([m_raf, m_oc], claims[:2], coeffs[:2], [0, 0], rounds, 3, label=6)
    for key in ("polys", "challenges", "member_claims", "final_claim"):
        assert np.array_equal(ref_ops[key], want[key]), key
    # the three-operator batch: its own consistency -- the final claim is the batching combination of the members' claims, and the operators' output claims are those of
    # the operators driven alone at the same point (the bound values are functions of the challenges only)
    acc = np.zeros(4, dtype=np.uint64)
    for c, mc in zip(coeffs, got["member_claims"]):
        acc = ffi.host_fr_add(acc, ffi.host_fr_mul(c, mc))
    assert np.array_equal(acc, got["final_claim"])
    f = folded_again(ctx, dops, raf)
    assert np.array_equal(ops[0].output_claims()[0], ctx.evaluate(f, got["challenges"][::-1]))
    f.free()
    for o in ops + two:
        o.destroy()
    for t in (init, val_final, val_io):
        t.free()
    m_raf.destroy()
    m_oc.destroy()
    index.free()
    dev.close()
    ctx.close()


def folded_again(ctx, dops, raf):
    eq = dops.eq(raf["tau_low"])
    folded = dops.pushforward("ram", [eq])[0]
    eq.free()
    return folded
