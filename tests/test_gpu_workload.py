"""GPU parity of the whole bench workload (SURVEY.md 8 a13 catalogue, every stage's batched sumcheck): the fused
LC-form device members with skipped s(1) must reproduce, bit for bit, the transcript of the oracle's naive flat-Expr
members -- the same claim the reference makes for its optimized tier vs its reference tier
(crates/jolt-kernels/src/optimized/parity.rs:79-118, tests/dory_byte_diff.rs)."""
import numpy as np
import pytest

from jolt_amd import ffi
from jolt_amd.workload import DeviceWorkload
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
