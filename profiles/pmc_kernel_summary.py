#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters of a rocprofv3 --pmc run (rocpd sqlite): one line per kernel name with every
counter summed over the dispatch's dimensions and averaged over the dispatches.  Usage: pmc_kernel_summary.py <results.db> [name-filter]"""
import collections
import sqlite3
import sys


def main(path, flt=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    did = "dispatch_id" if "dispatch_id" in cols else "id"
    vcol = "value" if "value" in cols else "counter_value"
    q = f"select {name_col}, {did}, counter_name, sum({vcol}) from counters_collection group by {name_col}, {did}, counter_name"
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, _, cname, v in cur.execute(q):
        if flt and flt not in name:
            continue
        per[name][cname].append(v)
    for name, cs in per.items():
        n = max(len(v) for v in cs.values())
        print(f"{name[:70]:70s} dispatches={n}")
        for cname, vals in sorted(cs.items()):
            print(f"    {cname:28s} avg={sum(vals) / len(vals):16.1f} max={max(vals):16.1f}")
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        # derived: share of the wavefronts' resident cycles in which a VALU instruction was executing / any instruction was executing / waiting
        # (SQ_ACTIVE_INST_* and SQ_WAIT_* count wave-cycles, as SQ_WAVE_CYCLES does; all are summed over the dispatch's dimensions)
        if avg.get("SQ_WAVE_CYCLES"):
            wc = avg["SQ_WAVE_CYCLES"]
            parts = []
            for label, key in (("VALU-active", "SQ_ACTIVE_INST_VALU"), ("any-inst-active", "SQ_ACTIVE_INST_ANY"), ("waiting", "SQ_WAIT_ANY"), ("waiting-on-inst", "SQ_WAIT_INST_ANY")):
                if key in avg:
                    parts.append(f"{label} {100.0 * avg[key] / wc:5.1f} %")
            if "SQ_INSTS_VALU" in avg and avg.get("SQ_ACTIVE_INST_VALU"):
                parts.append(f"cycles per VALU inst {avg['SQ_ACTIVE_INST_VALU'] / avg['SQ_INSTS_VALU']:4.1f}")
            print("    derived (of wave-cycles):    " + ",  ".join(parts))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
