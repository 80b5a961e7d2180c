"""GPU parity of the whole bench workload (SURVEY.md 8 a13 catalogue, every stage's batched sumcheck): the fused
LC-form device members with skipped s(1) must reproduce, bit for bit, the transcript of the oracle's naive flat-Expr
members -- the same claim the reference makes for its optimized tier vs its reference tier
(crates/jolt-kernels/src/optimized/parity.rs:79-118, tests/dory_byte_diff.rs)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from jolt_amd import ffi
from jolt_amd.workload import DeviceWorkload
from util import rand_challenge
from workload_oracle import OracleWorkload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_vars", [6, 9])
def test_catalogue_stage_transcripts_match_oracle(n_vars):
    ctx = ffi.Context(0)
    dev = DeviceWorkload(ctx, n_vars, seed=11)
    orc = OracleWorkload(n_vars, seed=11)
    want = orc.prove(label=100)
    for rep in range(2):  # second pass exercises reset()
        got = dev.prove(label=100)
        for stage in want:
            assert np.array_equal(got[stage]["polys"], want[stage]["polys"]), stage
            assert np.array_equal(got[stage]["challenges"], want[stage]["challenges"]), stage
            assert np.array_equal(got[stage]["final_claim"], want[stage]["final_claim"]), stage
    for i, c in enumerate(dev.claims):
        st = dev.members_spec[i].stage
        assert np.array_equal(c, want[st]["claims"][dev.stages[st].index(i)])
    ctx.close()


def test_catalogue_matches_oracle_at_benchmark_scale():
    """T = 2^20 (BASELINE configs[1] scale), the whole 11-relation catalogue: every round polynomial, challenge, final claim and input
    claim of the device path -- lazy one-hot members, eq-weighted members, linear-leaf fusions, skipped s(1), the grid-by-size rule and
    two-level tickets of the big rounds, tail kernels of the small ones -- equals the oracle's naive flat-Expr members over the dense
    tables (its sweeps run under OpenMP; crates/jolt-kernels/src/optimized/parity.rs:79-118 is the reference's form of this test)."""
    O.baseline_set_threads(min(48, os.cpu_count() or 1))  # the oracle's OpenMP sweeps: more threads than this only add merge overhead
    ctx = ffi.Context(0)
    dev = DeviceWorkload(ctx, 20)
    orc = OracleWorkload(20)
    want = orc.prove(label=300)
    got = dev.prove(label=300)
    for stage in want:
        assert np.array_equal(got[stage]["polys"], want[stage]["polys"]), stage
        assert np.array_equal(got[stage]["challenges"], want[stage]["challenges"]), stage
        assert np.array_equal(got[stage]["final_claim"], want[stage]["final_claim"]), stage
    for i, c in enumerate(dev.claims):
        st = dev.members_spec[i].stage
        assert np.array_equal(c, want[st]["claims"][dev.stages[st].index(i)])
    dev.close()
    ctx.close()


def test_stages_match_oracle_at_configs2_scale():
    """T = 2^22 (BASELINE configs[2] scale, the benchmarked size): ALL five stages of the 11-relation catalogue against the oracle,
    transcript for transcript (every round polynomial, challenge and final claim)."""
    O.baseline_set_threads(min(48, os.cpu_count() or 1))
    ctx = ffi.Context(0)
    dev = DeviceWorkload(ctx, 22)
    orc = OracleWorkload(22)
    want = orc.prove(label=500)
    got = dev.prove(label=500)
    for stage in want:
        for key in ("polys", "challenges", "final_claim"):
            assert np.array_equal(got[stage][key], want[stage][key]), (stage, key)
    dev.close()
    ctx.close()


def test_persistent_round_engine_matches_per_round_kernels(monkeypatch):
    """JOLT_ENGINE=1: the late rounds of every stage run inside the persistent round-engine kernel (engine_kernel.hip.h:
    fused binds, mailbox handshake per challenge).  All 11 relations, borrowed tables, expression / split-eq product /
    uniform members: transcripts must be bit-identical to the per-round kernels', also after a reset, and a caller that
    walks away mid-batch must not wedge the stream."""
    from jolt_amd.workload import DeviceWorkload
    base_ctx = ffi.Context(0)
    want = DeviceWorkload(base_ctx, 12).prove(label=77)
    base_ctx.close()
    for pairs in ("64", "1024"):
        monkeypatch.setenv("JOLT_ENGINE", "1")
        monkeypatch.setenv("JOLT_ENGINE_PAIRS", pairs)
        ctx = ffi.Context(0)  # the policy is read when the context first needs the engine
        wl = DeviceWorkload(ctx, 12)
        for rep in range(2):
            got = wl.prove(label=77)
            for stage in want:
                for k in ("polys", "challenges", "member_claims", "final_claim"):
                    assert np.array_equal(got[stage][k], want[stage][k]), (pairs, rep, stage, k)
        # abandon a batch while the engine is waiting for its next challenge: the next call quiesces it
        ms = [wl.members[i] for i in wl.stages[min(wl.stages)]]
        sums = ctx.round_group_prove(ms, [None] * len(ms))
        r = rand_challenge(5)
        for _ in range(9):  # 2^12 -> 2^3 entries: the engine has taken over for both thresholds and is mid-batch
            sums = ctx.round_group_prove(ms, [r] * len(ms))
        assert all(len(s) for s in sums)
        for m in ms:
            m.reset()
        again = wl.prove(label=77)
        for stage in want:
            assert np.array_equal(again[stage]["polys"], want[stage]["polys"]), (pairs, stage)
        ctx.close()


def test_large_trace_waits_for_side_stream_kernels():
    """T = 2^22 (BASELINE configs[2] scale): a round's kernels run for milliseconds on the side streams while the main stream is
    already idle; the completion wait must consult every stream before declaring a round lost (regression: 'batch round finished
    without publishing its completion flag').  No oracle at this size: the run must pass the prover's own round checks and be
    reproducible bit for bit."""
    ctx = ffi.Context(0)
    wl = DeviceWorkload(ctx, 22)
    a = wl.prove(label=5)
    b = wl.prove(label=5)
    for stage in a:
        assert np.array_equal(a[stage]["polys"], b[stage]["polys"]) and np.array_equal(a[stage]["final_claim"], b[stage]["final_claim"])
    ctx.close()


@pytest.mark.parametrize("knobs", [
    {"JOLT_GRID_MULT": "8"},          # 2048 workgroups per round kernel: the two-level completion tickets
    {"JOLT_LAZY_LDS": "0"},           # index-encoded members gather their branch tables from global memory
    {"JOLT_SERIAL_STREAMS": "1"},     # every kernel of a round on the main stream
    {"JOLT_TAIL_PAIRS": "256", "JOLT_FUSE_RATIO": "100"},  # group kernels (bind fused everywhere) down to small rounds
    {"JOLT_FUSE_TAIL": "1"},          # pending binds applied inside the tail kernel
    {"JOLT_UNIFORM_ROWS_PAIRS": "64", "JOLT_LAZY_LDS": "0"},  # one-item-per-pair forms of the uniform and lazy kernels
])
def test_alternate_kernel_paths_give_the_same_transcript(monkeypatch, knobs):
    """The measurement knobs of docs/multi_gpu.md section 6b select other kernels / grids / streams for the same sums: every one of them must
    reproduce the default path's transcript bit for bit (T = 2^16: above every switch-over threshold, two-level tickets included)."""
    base = ffi.Context(0)
    want = DeviceWorkload(base, 16, seed=5).prove(label=9)
    base.close()
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    ctx = ffi.Context(0)  # the knobs are read when the context is created
    got = DeviceWorkload(ctx, 16, seed=5).prove(label=9)
    ctx.close()
    for stage in want:
        for key in ("polys", "challenges", "final_claim"):
            assert np.array_equal(got[stage][key], want[stage][key]), (knobs, stage, key)


@pytest.mark.parametrize("n_vars,mode", [(6, True), (15, True), (12, "pinned"), (13, "overlapped")])  # "overlapped": the next proof's rows copied under this proof's kernels
def test_witness_upload_path_gives_the_same_proof(n_vars, mode):
    """DeviceWorkload(witness_upload=True): every step starts from the packed per-cycle rows in host memory -- one upload, the integer columns (jolt_ints_from_rows)
    and the RA chunk indices (jolt_onehot_from_rows: the nibbles of the instruction lookup index / the RAM address, cold = invalid) extracted on the device
    (RowSource::rows + WitnessBundle::from_row, crates/jolt-witness/src/consumer.rs:129-143) -- and proves what the resident witness proves"""
    ctx = ffi.Context(0)
    a = DeviceWorkload(ctx, n_vars, seed=12)
    b = DeviceWorkload(ctx, n_vars, seed=12, witness_upload=mode)  # "pinned": the row buffer in jolt_host_pinned_alloc memory
    rows, fields = b.pack_witness_rows()
    assert rows.shape == (1 << n_vars, b.witness_bytes_per_cycle()) and len(fields) == len(b.ints) + len(b.sources)
    want = a.prove(label=7)
    for rep in range(3 if mode == "overlapped" else 2):  # (overlapped: the first proof waits for its own copy, the later ones for a copy begun a proof earlier)
        got = b.step(label=7)["stages"]
        for stage in want:
            for key in ("polys", "challenges", "final_claim"):
                assert np.array_equal(got[stage][key], want[stage][key]), (rep, stage, key)
    for i in b.sources:
        assert np.array_equal(b.sources[i].download(), a.sources[i].download())
    a.close()
    b.close()
    ctx.close()


def test_rows_handle_is_refused_before_its_copy_has_been_waited_for():
    """jolt_rows_upload_begin's handle is pending until jolt_rows_upload_wait: the extraction kernels run on the main stream, which does not depend on the copy
    stream, so every *_from_rows entry point must refuse it (JOLT_ERR_INVALID_ARG) rather than extract from a partly copied block -- and serve it after the wait"""
    ctx = ffi.Context(0)
    T, row_bytes = 1 << 10, 24
    buf = ffi.PinnedBuffer(ctx, (T, row_bytes))
    rng = np.random.default_rng(3)
    buf.array[...] = rng.integers(0, 256, size=(T, row_bytes), dtype=np.uint8)
    rows = ffi.Rows.begin(ctx, buf.array)
    calls = [lambda: rows.table(0, 8), lambda: rows.ints(8, 8), lambda: rows.ints_many([(0, 8, False), (8, 8, True)]), lambda: rows.onehot(16, 2, [0, 4, 8, 12], 4, valid_offset=18),
             lambda: rows.window_table(0, 8, False, 1, T), lambda: rows.onehot_sentinel(16, 2, [0, 4], 4, T)]
    for call in calls:
        with pytest.raises(ffi.JoltError) as e:
            call()
        assert e.value.status == 1 and "jolt_rows_upload_wait" in str(e.value), str(e.value)  # JOLT_ERR_INVALID_ARG
    rows.wait()
    col = rows.ints(8, 8)
    want = ctx.from_u64(buf.array[:, 8:16].copy().view(np.uint64).reshape(T))
    assert np.array_equal(ctx.table_from_ints(col).download(), want.download())
    many = rows.ints_many([(0, 8, False), (8, 8, False)])
    assert np.array_equal(ctx.table_from_ints(many[1]).download(), want.download())
    for v in many + [col]:
        v.free()
    rows.free()
    buf.free()
    ctx.close()


@pytest.mark.parametrize("n_vars,contexts", [(9, "2"), (16, "2"), (12, "1")])
def test_concurrent_stages_prove_what_the_serial_order_proves(n_vars, contexts, monkeypatch):
    """DeviceWorkload.step runs, per protocol stage, the stage operators on their own contexts and host threads beside the stage's batched sumcheck (prove_stages): every
    operator's transcript and claims and every catalogue stage must be what the operators one after the other on ONE context (DeviceExtended.prove) and the catalogue
    alone (DeviceWorkload.prove) produce -- three times over, so that a race between the contexts would have three chances to show"""
    from jolt_amd.stages import DeviceExtended
    from test_gpu_extended import same
    monkeypatch.setenv("JOLT_STAGE_CONTEXTS", contexts)
    ctx = ffi.Context(0)
    serial_ext = DeviceExtended(ctx, n_vars, seed=31)
    want_ext = serial_ext.prove(label=ffi.TRANSCRIPT_BLAKE2B | 60)
    serial_ext.close()
    plain = DeviceWorkload(ctx, n_vars, seed=31)
    want = plain.prove(label=ffi.TRANSCRIPT_BLAKE2B | 60)
    plain.close()
    wl = DeviceWorkload(ctx, n_vars, seed=31, extended=True)
    assert wl._stage_worker is not None and len(wl._stage_slots) == int(contexts)
    for rep in range(3):
        out = wl.step(label=ffi.TRANSCRIPT_BLAKE2B | 60)
        assert set(out["extended"]) == set(want_ext)
        for name in want_ext:
            same(out["extended"][name], want_ext[name], f"rep {rep} {name}")
        for stage in want:
            for key in ("polys", "challenges", "final_claim"):
                assert np.array_equal(out["stages"][stage][key], want[stage][key]), (rep, stage, key)
    wl.close()
    monkeypatch.setenv("JOLT_STAGE_CONCURRENCY", "0")
    serial = DeviceWorkload(ctx, n_vars, seed=31, extended=True)
    assert serial._stage_worker is None and serial.ext_ctx is None
    out = serial.step(label=ffi.TRANSCRIPT_BLAKE2B | 60)
    for name in want_ext:
        same(out["extended"][name], want_ext[name], f"serial {name}")
    serial.close()
    ctx.close()
