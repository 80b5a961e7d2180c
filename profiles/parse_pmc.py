#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of profiles/collect_bind_traffic.sh into profiles/bind_traffic.json.

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch (TCC_EA0 request counters x 64 B / 1024).  Correction per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-byte fabric requests at 64 B, i.e.
reports exactly half of a wide (16 B/lane) streaming read -- it is doubled here; WRITE_SIZE is uncalibrated by the guide
and is reported as measured, next to the algorithmic byte counts so the calibration is visible."""
import glob
import json
import sqlite3
import sys


def per_dispatch(dbdir, counter, kernel_like):
    vals = []
    for db in glob.glob(f"{dbdir}/**/*_results.db", recursive=True):
        con = sqlite3.connect(db)
        cur = con.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tabs:
            continue
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
        cname = "counter_name" if "counter_name" in cols else None
        vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
        if not (name_col and cname and vcol):
            print(json.dumps({"error": "unexpected counters_collection schema", "columns": cols}))
            return vals
        did = "dispatch_id" if "dispatch_id" in cols else "id"
        q = f"select {did}, sum({vcol}) from counters_collection where {cname} = ? and {name_col} like ? group by {did}"
        vals += [r[1] for r in cur.execute(q, (counter, f"%{kernel_like}%"))]
    return vals


def main(out):
    kernel = "k_bind_low_to_high"
    f = per_dispatch(f"{out}/FETCH_SIZE", "FETCH_SIZE", kernel)
    w = per_dispatch(f"{out}/WRITE_SIZE", "WRITE_SIZE", kernel)
    bench = json.load(open(f"{out}/FETCH_SIZE.json"))
    n = bench["table_len"]
    res = {"kernel": kernel, "table_len": n, "dispatches_fetch": len(f), "dispatches_write": len(w),
           "algorithmic_read_bytes": 32 * n, "algorithmic_write_bytes": 16 * n}
    if f and w:
        fk = sorted(f)[len(f) // 2]
        wk = sorted(w)[len(w) // 2]
        res.update({"FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk,
                    "read_bytes_corrected": 2 * fk * 1024, "write_bytes": wk * 1024,
                    "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024,
                    "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as measured"})
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1])
