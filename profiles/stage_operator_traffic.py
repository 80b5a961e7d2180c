#!/usr/bin/env python3
"""Join three rocprofv3 passes over the same command (tools/pmc_extended.sh: --kernel-trace alone, --pmc FETCH_SIZE, --pmc WRITE_SIZE) into one per-kernel table:
calls, total time (from the counter-free pass), HBM bytes read / written per PROOF and the rate they imply.

FETCH_SIZE / WRITE_SIZE come in KiB per dispatch; per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 128-byte requests at 64 B
and is doubled here, WRITE_SIZE is taken as measured.  Usage: stage_operator_traffic.py <dir with trace/ FETCH_SIZE/ WRITE_SIZE/> <proofs per run>"""
import collections
import glob
import sqlite3
import sys


def db_of(d):
    f = glob.glob(f"{d}/**/*_results.db", recursive=True)
    return sqlite3.connect(f[0]) if f else None


def counter_totals(d, counter):
    con = db_of(d)
    out = collections.defaultdict(float)
    if con is None:
        return out
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    vcol = "value" if "value" in cols else "counter_value"
    for name, v in cur.execute(f"select {name_col}, sum({vcol}) from counters_collection where counter_name = ? group by {name_col}", (counter,)):
        out[name] += v
    return out


def main(root, proofs):
    # the run makes one warm-up proof + `proofs` timed ones: every total is divided by (proofs + 1)
    runs = proofs + 1
    con = db_of(f"{root}/trace")
    rows = list(con.cursor().execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc"))
    fetch = counter_totals(f"{root}/FETCH_SIZE", "FETCH_SIZE")
    write = counter_totals(f"{root}/WRITE_SIZE", "WRITE_SIZE")
    print(f"# per proof of the stage operators ({runs} proofs in the run, the resident-input builds of DeviceExtended.__init__ included in the first)")
    print(f"{'kernel':84s} {'calls':>7s} {'ms':>8s} {'read_MB':>9s} {'write_MB':>9s} {'GB/s':>8s}")
    tot_t = tot_r = tot_w = 0.0
    for name, n, total in rows:
        t = total / 1e6 / runs
        r = 2.0 * fetch.get(name, 0.0) * 1024 / 1e6 / runs
        w = write.get(name, 0.0) * 1024 / 1e6 / runs
        tot_t, tot_r, tot_w = tot_t + t, tot_r + r, tot_w + w
        if t < 0.02:
            continue
        print(f"{name[:84]:84s} {n / runs:7.1f} {t:8.3f} {r:9.1f} {w:9.1f} {(r + w) / t if t else 0:8.1f}")
    print(f"{'TOTAL':84s} {'':7s} {tot_t:8.3f} {tot_r:9.1f} {tot_w:9.1f} {(tot_r + tot_w) / tot_t:8.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
