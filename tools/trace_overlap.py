"""Kernel-trace timeline questions on a rocprofv3 rocpd database: does kernel family X run concurrently with family Y, and how
long do Y's launches take while an X launch is in flight?  Usage: python tools/trace_overlap.py <results.db> <X substr> <Y substr>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else None
rows = cur.execute(f"select s.kernel_name, d.start, d.end{', d.' + qcol if qcol else ''} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
X, Y = sys.argv[2], sys.argv[3]
xs = [(r[1], r[2]) for r in rows if X in r[0]]
ys = [(r[1], r[2], r[0]) for r in rows if Y in r[0]]
print(f"{len(rows)} dispatches, {len(xs)} x '{X}', {len(ys)} x '{Y}', queues: {sorted({r[3] for r in rows}) if qcol else 'n/a'}")
if qcol:
    print(" queues used by X:", sorted({r[3] for r in rows if X in r[0]}), " by Y:", sorted({r[3] for r in rows if Y in r[0]}))
import bisect
xstarts = [a for a, _ in xs]
ov_tot, y_ov, y_free = 0, [], []
for s, e, n in ys:
    i = bisect.bisect_left(xstarts, s) - 1
    ov = 0
    for a, b in xs[max(i, 0):]:
        if a >= e:
            break
        ov += max(0, min(e, b) - max(s, a))
    ov_tot += ov
    (y_ov if ov > 0.5 * (e - s) else y_free).append(e - s)
xt = sum(b - a for a, b in xs)
print(f"X total {xt/1e6:.2f} ms, of which overlapped with Y {ov_tot/1e6:.2f} ms")
for lab, v in (("Y launches mostly overlapped by X", y_ov), ("Y launches free of X", y_free)):
    if v:
        print(f"  {lab}: n={len(v)} avg {sum(v)/len(v)/1e3:.1f} us")
# busy union vs span of the last 40 % of the trace (steady state)
t0 = rows[int(len(rows) * 0.6)][1]
iv = sorted((max(s, t0), e) for _, s, e, *_ in rows if e > t0)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"steady-state window {(iv[-1][1]-t0)/1e6:.1f} ms: GPU busy (union) {busy/1e6:.1f} ms, sum of kernel durations {sum(e-s for s,e in iv)/1e6:.1f} ms")
