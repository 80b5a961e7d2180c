import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# second open = second half: split at the largest idle gap in the middle? use last 45% of launches by time
t_end = rows[-1][2]
# find start of second open: first k_bind/hyperkzg fold after the midpoint... approximate: take kernels in the last `dur` where dur from argv
dur = float(sys.argv[2]) * 1e6
sel = [r for r in rows if r[1] >= t_end - dur]
ev = sorted([(r[1], 1) for r in sel] + [(r[2], -1) for r in sel])
busy = 0; depth = 0; last = None
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print(f"window {dur/1e6:.1f} ms: busy {busy/1e6:.2f} ms, kernels {len(sel)}")
agg = {}
for n, s, e in sel:
    m = re.search(r"k_[a-z0-9_]+", n); k = m.group(0) if m else n[:30]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:32s} {c:5d} calls {t/1e6:8.2f} ms")
