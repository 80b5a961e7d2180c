#!/usr/bin/env python3
"""Where the time of jolt_amd/stages.py's operators goes (a synchronisation after every part): time_extended.py [log_t]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jolt_amd import ffi  # noqa: E402
from jolt_amd import stages as S  # noqa: E402


def main():
    log_t = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    ctx = ffi.Context(0)
    e = S.DeviceExtended(ctx, log_t)
    d, lk, rr = e.d, e.d["lookup"], e.read_raf
    for rep in range(2):
        acc = {}

        def lap(name, t0):
            ctx.synchronize()
            acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

        tr = ffi.HostTranscript(5)
        t0 = time.perf_counter(); u = ctx.eq_evals(d["lookup_reduction"]); lap("eq", t0)
        present = np.zeros(S.N_LOOKUP_TABLES, dtype=np.uint8)
        present[lk["present"]] = 1
        t0 = time.perf_counter(); state = ffi.HostReadRafAddress(d["lookup_gamma"], present); lap("host_create", t0)
        claim = e.claims["lookup"]
        v_tables = []
        for phase in range(S.PHASES):
            suffix_len = S.ADDRESS_BITS - 8 * (phase + 1)
            if phase:
                t0 = time.perf_counter(); rr.condense(u, v_tables[-1], suffix_len + 8); lap("condense", t0)
            t0 = time.perf_counter(); raf, suf = rr.phase_scan(u, suffix_len, S.ADDRESS_BITS, e.lookup_lists); lap("scan", t0)
            t0 = time.perf_counter(); state.init_phase(phase, raf, suf); lap("host_init_phase", t0)
            if claim is None:
                m = state.message()
                claim = e.claims["lookup"] = ffi.host_fr_add(m[0], m[1])
            t0 = time.perf_counter(); claim, _, _ = state.prove_phase(claim, tr); lap("host_8_rounds", t0)
            t0 = time.perf_counter(); v_tables.append(state.v_table(phase)); lap("host_v_table", t0)
        u.free()
        vt = np.stack(v_tables)
        t0 = time.perf_counter(); tv, ri, rid = state.finish(); state.close(); lap("host_finish", t0)
        t0 = time.perf_counter(); combined, ra = rr.cycle_tables(tv, ri, rid, vt, S.ADDRESS_BITS, d["ra_count"]); lap("cycle_tables", t0)
        n_f = 1 + d["ra_count"]
        t0 = time.perf_counter(); member = ctx.member_lc([combined] + ra, [[(None, [(e.one, i)]) for i in range(n_f)]], n_f, eq_point=d["lookup_reduction"]); lap("member", t0)
        t0 = time.perf_counter(); ctx.prove_batch([member], [claim], [e.one], [0], log_t, n_f + 1, label=6); lap("cycle_rounds", t0)
        member.destroy()
        print({k: round(v, 2) for k, v in acc.items()}, flush=True)
    for name, fn in (("spartan_outer", lambda: e.spartan(e.outer_ints, d["outer_iwa"], d["outer_iwb"], d["outer_wa"], d["outer_wb"], d["outer_tau"], d["outer_kernel"], e.claims["outer"], 2, 7)),
                     ("ram", lambda: e.ram_read_write(8)), ("registers", lambda: e.registers_read_write(9)), ("booleanity_address", lambda: e.booleanity_address(10))):
        fn(); ctx.synchronize(); t0 = time.perf_counter(); fn(); ctx.synchronize()
        print(name, round((time.perf_counter() - t0) * 1e3, 2), "ms")
    # the address-domain relations part by part: index builds, then each driver over prebuilt indexes
    ram, bc = d["ram"], d["bytecode"]
    for rep in range(2):
        acc = {}
        def timed(name, fn):
            ctx.synchronize(); t0 = time.perf_counter(); r = fn(); ctx.synchronize(); acc[name] = round((time.perf_counter() - t0) * 1e3, 2); return r
        pc_ix = timed("pc_index", lambda: ctx.key_index(e.pc_ints, 1 << bc["log_k"]))
        ram_ix = timed("ram_index", lambda: ctx.key_index(e.ram_cols[0], 1 << ram["log_k"]))
        ops = S.DeviceOps(ctx, ffi, {"pc": pc_ix, "ram": ram_ix, "ram_post": e.ram_cols[2]}, e.pc_chunks)
        timed("bytecode_read_raf", lambda: S.bytecode_read_raf(ops, bc, log_t, 20))
        timed("ram_raf_evaluation", lambda: S.ram_raf_evaluation(ops, ram, d["ram_raf"], 30))
        timed("ram_output_check", lambda: S.ram_output_check(ops, ram, d["ram_output"], 40))
        pc_ix.free(); ram_ix.free()
        print("address_domain", acc, "items", flush=True)


if __name__ == "__main__":
    main()
