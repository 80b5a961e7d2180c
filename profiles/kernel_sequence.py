#!/usr/bin/env python3
"""Kernel launches of a rocprofv3 --kernel-trace results db in time order, one line each (short name, grid, duration in us);
with a second argument N only the last N launches.  Used for the per-call breakdowns quoted in DESIGN.md."""
import re
import sqlite3
import sys


def main(path, last=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    if last:
        rows = rows[-int(last):]
    t0 = rows[0][1] if rows else 0
    for name, st, en, gx, gy, gz in rows:
        m = re.search(r"k_[a-z0-9_]+", name)
        print(f"{(st - t0) / 1e3:10.0f} us  {m.group(0) if m else name[:40]:32s} grid=({gx},{gy},{gz}) {(en - st) / 1e3:9.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
