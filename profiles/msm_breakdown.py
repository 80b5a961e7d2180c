#!/usr/bin/env python3
"""Per-MSM kernel breakdown from a rocprofv3 --kernel-trace results db: one line per jolt_msm call (digits ... window_reduce)."""
import re
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    seq = None
    for name, st, en, gx in rows:
        m = re.search(r"k_msm_[a-z_]+", name)
        if not m:
            continue
        k = m.group(0)[6:]
        if k == "digits":
            seq = {"n": gx, "t0": st, "parts": []}
        if seq is None:
            continue
        seq["parts"].append((k, (en - st) / 1e3))
        if k == "window_reduce":
            parts = " ".join(f"{k}={t:.0f}" for k, t in seq["parts"])
            print(f"n<={seq['n']:8d} total_us={(en - seq['t0']) / 1e3:8.0f}  {parts}")
            seq = None


if __name__ == "__main__":
    main(sys.argv[1])
