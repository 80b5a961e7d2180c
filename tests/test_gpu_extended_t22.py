"""The stage 1 / 2 / 4 / 5 / 6a / 6b / 7 operators at the BENCHMARKED size, T = 2^22 cycles (bench.py's default step runs them at exactly this size: K = 2^16 RAM words,
42 lookup tables, 35 R1CS inputs, 36 RA columns, a 2^12-entry bytecode), one test per operator so that each fits the per-test time limit.  The sizes differ from the
T = 2^20 run of tests/test_gpu_extended.py in more than length: read_raf.hip cuts 4x as many 1024-row work items per bin, key_index.hip's 32768-bin passes hold 4x the
rows per bin, the round kernels take the two-level ticket path with the large grids, the sparse matrices merge 4x as many cells per group.

Device (jolt_amd.stages.DeviceExtended over libjolt_hip.so) against tests/workload_oracle.py:OracleExtended, the oracle's OpenMP sweeps, message for message
(crates/jolt-kernels/src/optimized/parity.rs:79-118 is the reference's form of this test).  What "against the oracle" means for the 128 address rounds of instruction
read-RAF at this size is spelled out in OracleExtended.instruction_read_raf: rounds {0, 62, 127} are computed FROM THE DEFINITION by the oracle (evaluate_mle of every
row's table) and must equal the product's; the other rounds are checked as a sumcheck from independent ends (first-principles input claim, s(0) + s(1) = claim every
round, the oracle's evaluate_mle at r_address at the end) -- sumcheck-ends-verified, not lock step."""
import os

import numpy as np
import pytest

from jolt_amd import ffi
from jolt_amd.stages import DeviceExtended, build_extended
from test_gpu_extended import same
from workload_oracle import OracleExtended

pytestmark = pytest.mark.gpu

N_VARS, SEED, LABEL = 22, 2026, 40


@pytest.fixture(scope="module")
def pair():
    import oracle_lib as O
    O.baseline_set_threads(min(128, os.cpu_count() or 1))
    d = build_extended(N_VARS, SEED)  # one description (numpy, ~20 s) shared by both sides
    ctx = ffi.Context(0)
    dev = DeviceExtended(ctx, N_VARS, description=d)
    orc = OracleExtended(N_VARS, description=d)
    yield dev, orc
    dev.close()
    ctx.close()


def check(dev, got, want, name, claim_key=None):
    if claim_key is not None:  # the input claim the device side computed once from its resident columns against the oracle's dense definition
        assert np.array_equal(dev.claims[claim_key], want["claim"]), name
        want = {k: v for k, v in want.items() if k != "claim"}
    same(got, want, name)


def test_spartan_outer_at_benchmark_scale(pair):
    dev, orc = pair
    d = dev.d
    got = dev.spartan(dev.outer_ints, d["outer_iwa"], d["outer_iwb"], d["outer_wa"], d["outer_wb"], d["outer_tau"], d["outer_kernel"], dev.claims["outer"], 2, LABEL + 100)
    check(dev, got, orc.spartan_outer(LABEL + 100), "spartan_outer", "outer")


def test_spartan_product_at_benchmark_scale(pair):
    dev, orc = pair
    d = dev.d
    got = dev.spartan(dev.product_ints, dev.product_ia, dev.product_ib, dev.product_fa, dev.product_fb, d["product_tau"], d["product_kernel"], dev.claims["product"], 1, LABEL + 200)
    check(dev, got, orc.spartan_product(LABEL + 200), "spartan_product", "product")


def test_ram_read_write_at_benchmark_scale(pair):
    dev, orc = pair
    check(dev, dev.ram_read_write(LABEL + 300), orc.ram_read_write(LABEL + 300), "ram_read_write", "ram")


def test_registers_read_write_at_benchmark_scale(pair):
    dev, orc = pair
    check(dev, dev.registers_read_write(LABEL + 350), orc.registers_read_write(LABEL + 350), "registers_read_write", "registers")


def test_instruction_read_raf_at_benchmark_scale(pair):
    """16 phase scans (sum for sum), 128 address rounds (4 of them from the definition, see the module docstring), cycle columns, 22 cycle rounds, output claims"""
    dev, orc = pair
    got = dev.instruction_read_raf(LABEL + 400)
    want = orc.instruction_read_raf(LABEL + 400)
    assert orc.direct_checked == sorted(OracleExtended.sampled_direct_rounds(N_VARS)) and len(orc.direct_checked) >= 3
    check(dev, got, want, "instruction_read_raf", "lookup")


def test_pushforward_operators_at_benchmark_scale(pair):
    """booleanity address phase (stage 6a) and Hamming-weight claim reduction (stage 7): the 36-column pushforwards at T = 2^22 + their K-domain rounds"""
    dev, orc = pair
    got, want = dev.booleanity_address(LABEL + 450), orc.booleanity_address(LABEL + 450)
    same(got, {k: v for k, v in want.items() if k != "claim"}, "booleanity_address")
    same(dev.hamming_weight(LABEL + 470), orc.hamming_weight(LABEL + 470), "hamming_weight")
    # stage 6b: the booleanity cycle phase over the same 36 lazily bound columns, 22 rounds, from the address phase's bound point
    r_address = got["challenges"][::-1]
    cyc, cyc_want = dev.booleanity_cycle(LABEL + 460, r_address, got["intermediate"]), orc.booleanity_cycle(LABEL + 460, r_address)
    same(cyc, cyc_want, "booleanity_cycle")
    assert np.array_equal(cyc["claim"], got["intermediate"])


def test_address_domain_relations_at_benchmark_scale(pair):
    """bytecode read+RAF (address and cycle phases), RAM RAF evaluation, RAM output check over the key indexes of the PC and RAM address columns"""
    dev, orc = pair
    got, want = dev.address_domain(LABEL + 500), orc.address_domain(LABEL + 500)
    assert set(got) == set(want) == {"bytecode_read_raf", "ram_raf_evaluation", "ram_output_check"}
    for name in got:
        same(got[name], want[name], name)


def test_memory_checking_operators_on_a_hot_set_trace_at_benchmark_scale():
    """BASELINE configs[4]'s shape at T = 2^22 (`bench.py --ram-addresses hotset`): the RAM and register address streams skewed onto a hot set (90 % of the accesses on 2^10
    of the 2^16 words / 8 of the 128 registers, jolt_amd.stages.hotset_addresses) where every other test draws them uniformly.  The operators whose data structures
    depend on the address distribution -- both sparse read-write matrices, the RAM key index behind RAF evaluation and the output check -- against the oracle twin."""
    import oracle_lib as O
    O.baseline_set_threads(min(128, os.cpu_count() or 1))
    d = build_extended(N_VARS, SEED + 1, ram_addresses="hotset")
    hit = d["ram"]["addresses"][d["ram"]["addresses"] != np.uint64(0xFFFFFFFFFFFFFFFF)]
    assert np.unique(hit, return_counts=True)[1].max() > 1000  # (uniform: ~38 accesses per word)
    ctx = ffi.Context(0)
    dev = DeviceExtended(ctx, N_VARS, description=d)
    orc = OracleExtended(N_VARS, description=d)
    check(dev, dev.ram_read_write(LABEL + 300), orc.ram_read_write(LABEL + 300), "ram_read_write (hot set)", "ram")
    check(dev, dev.registers_read_write(LABEL + 350), orc.registers_read_write(LABEL + 350), "registers_read_write (hot set)", "registers")
    got, want = dev.address_domain(LABEL + 500), orc.address_domain(LABEL + 500)
    for name in ("ram_raf_evaluation", "ram_output_check"):
        same(got[name], want[name], name + " (hot set)")
    dev.close()
    ctx.close()
