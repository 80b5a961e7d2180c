# rocprofv3 --kernel-trace of bench.py (1 warm-up + 3 timed passes), per-kernel stats into gpurun_out/ and the kernel
# timeline of stage 6b of the last pass on stdout.  SERIAL=1: all kernels of a round on one stream (standalone durations).
cd /tmp && export TMPDIR=/tmp
JOLT_SERIAL_STREAMS=${SERIAL:-0} timeout 200 rocprofv3 --kernel-trace -d /tmp/prof -o b -- python /root/repo/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.txt 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
[ -z "$f" ] && { echo nodb; exit 1; }
timeout 60 python /root/repo/profiles/summarize_rocprof.py "$f" > /root/repo/gpurun_out/bench_v9_kernel_stats.txt
timeout 60 python - "$f" <<'PY'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "lazy_first" in r[0]]
first = idx[-2] - 3   # instruction_ra_virtualization / ram_ra of the last step: stage 6 starts here
t0 = rows[first][1]; prev = t0
for i in range(first, min(len(rows), first + 70)):
    n, s, e, g = rows[i]
    m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", n)
    print(f"{i:6d} t={(s-t0)/1e3:8.1f} dur={(e-s)/1e3:7.1f} gap={(s-prev)/1e3:6.1f} grid={g:8d} {m.group(0) if m else n[:30]}")
    prev = max(prev, e)
PY
