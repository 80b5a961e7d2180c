#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results database (rocpd sqlite) into the per-kernel table that `--stats`
prints: calls, total / average / min / max duration and share of GPU time, plus the busy/idle split of the traced
window.  Usage: python profiles/summarize_rocprof.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                            "group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    span = list(cur.execute("select min(start), max(end) from kernels"))[0]
    print(f"# source: {path}")
    print(f"# kernels traced: {sum(r[1] for r in rows)}  total kernel time: {tot/1e6:.3f} ms  traced window: {(span[1]-span[0])/1e6:.3f} ms")
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}")
    for name, n, total, avg, mn, mx in rows:
        print(f"{name[:90]:90s} {n:7d} {total/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:10.2f} {100*total/tot:6.2f}")
    try:
        regs = list(cur.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                                "from kernels group by name order by sum(end-start) desc"))
        print("\n# resources (arch VGPR, accum VGPR, SGPR, LDS bytes, scratch bytes)")
        for r in regs:
            print(f"{r[0][:90]:90s} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:7d} {r[5]:6d}")
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
