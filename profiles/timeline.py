#!/usr/bin/env python3
"""Print a slice of the kernel timeline of a rocprofv3 --kernel-trace results db: start offset, duration, gap to the previous
kernel end, grid size, short kernel name.  Usage: timeline.py <results.db> <first_index> <count> [name-filter]"""
import re
import sqlite3
import sys


def main(path, first, count, flt=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute("select name, start, end, grid_size_x from kernels order by start").fetchall() if _has(cur, "grid_size_x") else \
        [(n, s, e, 0) for n, s, e in cur.execute("select name, start, end from kernels order by start")]
    if flt:
        idx = [i for i, r in enumerate(rows) if flt in r[0]]
        print(f"# {len(idx)} kernels match {flt!r}; first at {idx[:5]}")
    t0 = rows[first][1]
    prev_end = rows[first - 1][2] if first else t0
    for i in range(first, min(len(rows), first + count)):
        name, st, en, gx = rows[i]
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", name)
        short = m.group(0) if m else name[:40]
        print(f"{i:6d} t={(st - t0) / 1e3:9.1f}us dur={(en - st) / 1e3:8.1f} gap={(st - prev_end) / 1e3:8.1f} grid={gx:9d} {short}")
        prev_end = max(prev_end, en)


def _has(cur, col):
    try:
        cur.execute(f"select {col} from kernels limit 1")
        return True
    except sqlite3.Error:
        return False


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
