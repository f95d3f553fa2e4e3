"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table (like --stats CSV).
Usage: python tools/rocpd_stats.py <results.db> [skip_first_n_dispatches_fraction]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
t0, t1 = rows[0][1], rows[-1][2]
stats = {}
for name, s, e in rows:
    n = name.split("(")[0]
    st = stats.setdefault(n, [0, 0, 1 << 62, 0])
    st[0] += 1; st[1] += e - s; st[2] = min(st[2], e - s); st[3] = max(st[3], e - s)
tot = sum(v[1] for v in stats.values())
print(f"# dispatches={len(rows)} kernel_time_total_ms={tot/1e6:.1f} wall_span_ms={(t1-t0)/1e6:.1f}")
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for n, (c, t, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:70]:70s} {c:7d} {t/1e6:10.2f} {t/c/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {100*t/tot:6.2f}")
