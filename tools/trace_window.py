"""List every dispatch inside the time window spanned by the n-th burst of kernels matching <substr> (rocprofv3 rocpd database).
Usage: python tools/trace_window.py <results.db> <substr> [burst_index] [max_rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
sub = sys.argv[2]
idx = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mx = int(sys.argv[4]) if len(sys.argv) > 4 else 80
hits = [(s, e) for n, s, e, q in rows if sub in n]
bursts, cur_b = [], [hits[0]]
for h in hits[1:]:
    if h[0] - cur_b[-1][1] > 50e6:
        bursts.append(cur_b)
        cur_b = [h]
    else:
        cur_b.append(h)
bursts.append(cur_b)
b = bursts[min(idx, len(bursts) - 1)]
w0, w1 = b[0][0] - 2e6, b[-1][1] + 2e6
print(f"{len(bursts)} bursts; burst {idx}: {len(b)} launches spanning {(b[-1][1]-b[0][0])/1e6:.2f} ms")
n = 0
for name, s, e, q in rows:
    if e < w0 or s > w1:
        continue
    print(f"q{q} {(s-w0)/1e3:10.1f} us +{(e-s)/1e3:8.1f}  {name.split('(')[0][-60:]}")
    n += 1
    if n >= mx:
        break
