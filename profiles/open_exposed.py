#!/usr/bin/env python3
"""What an opening waits for when it is NOT summing buckets.  Reads a rocprofv3 --kernel-trace results db of tools/open_one.py (two openings separated by a 50 ms sleep),
takes the LAST burst of kernels (split at idle gaps > 20 ms) and prints: the burst's length, the time during which at least one bucket-sum kernel runs
(k_fx_buckets_ordered, k_fx_heavy_segments, k_msm_buckets_*: the multiply-add bound floor), and the rest attributed to the kernels that run while no bucket sum does
(shared equally when several overlap) and to idle time (no kernel at all: launch gaps, host work, event waits).  Usage: open_exposed.py <results.db> [top]"""
import collections
import re
import sqlite3
import sys

BUCKET = ("k_fx_buckets_ordered", "k_fx_buckets_ordered_staged", "k_fx_heavy_segments", "k_fx_heavy_segments_staged", "k_msm_buckets_light", "k_msm_buckets_heavy")


def short(name):
    m = re.search(r"k_[a-z0-9_]+", name)
    return m.group(0) if m else name[:40]


def main(path, top=25):
    cur = sqlite3.connect(path).cursor()
    rows = [(short(n), s, e) for n, s, e in cur.execute("select name, start, end from kernels order by start")]
    # last burst
    first, reach = 0, rows[0][2]
    for i in range(1, len(rows)):
        if rows[i][1] - reach > 20e6:
            first = i
        reach = max(reach, rows[i][2])
    burst = rows[first:]
    t0, t1 = burst[0][1], max(r[2] for r in burst)
    ev = []
    for k, (n, s, e) in enumerate(burst):
        ev.append((s, 1, k))
        ev.append((e, 0, k))
    ev.sort()
    running, n_bucket = set(), 0
    bucket_busy = idle = 0.0
    exposed = collections.defaultdict(float)
    last = t0
    for t, kind, k in ev:
        dt = t - last
        if dt > 0:
            if n_bucket:
                bucket_busy += dt
            elif running:
                for j in running:
                    exposed[burst[j][0]] += dt / len(running)
            else:
                idle += dt
        last = t
        is_b = burst[k][0] in BUCKET
        if kind:
            running.add(k)
            n_bucket += is_b
        else:
            running.discard(k)
            n_bucket -= is_b
    total = t1 - t0
    sums = collections.defaultdict(float)
    for n, s, e in burst:
        sums[n] += e - s
    print(f"# {path}")
    print(f"burst: {len(burst)} kernels, {total / 1e6:.2f} ms;  a bucket-sum kernel running: {bucket_busy / 1e6:.2f} ms;  other kernels only: {sum(exposed.values()) / 1e6:.2f} ms;  idle: {idle / 1e6:.2f} ms")
    print(f"sum of bucket-sum kernel durations (overlapping lanes counted once each): {sum(v for k, v in sums.items() if k in BUCKET) / 1e6:.2f} ms")
    print(f"{'kernel running while no bucket sum runs':44s} {'exposed_ms':>10s} {'own_total_ms':>12s}")
    for n, v in sorted(exposed.items(), key=lambda kv: -kv[1])[:top]:
        print(f"{n:44s} {v / 1e6:10.3f} {sums[n] / 1e6:12.3f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
