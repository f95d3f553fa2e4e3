"""The front end of ONE traced training step, launch by launch: every kernel from the end of the previous optimizer burst to the first
decoder-layer attention, with its start offset, duration and the gap in front of it (rocprofv3 rocpd trace, as tools/step_breakdown.py).
Shows what the encoders / splice / host synchronisations cost before the first 7B GEMM and what could overlap.
Usage: python tools/step_frontend_timeline.py <results.db> [step_index_from_end=1]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
gx = "d.grid_size_x" if "grid_size_x" in cols else ("d.grid_x" if "grid_x" in cols else "0")
wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else ("d.workgroup_x" if "workgroup_x" in cols else "1")
qid = "d.queue_id" if "queue_id" in cols else "0"
rows = cur.execute(f"select s.kernel_name, d.start, d.end, {gx}, {wx}, {qid} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
opt = [(s, e) for n, s, e, *_ in rows if "adamw" in n]
bursts, cb = [], [opt[0]]
for h in opt[1:]:
    if h[0] - cb[-1][1] > 5e6:
        bursts.append(cb)
        cb = [h]
    else:
        cb.append(h)
bursts.append(cb)
t0 = bursts[-back - 1][-1][1]
last = t0
tot = gaps = 0.0
print(f"{'at us':>9} {'gap us':>8} {'dur us':>8} {'blocks':>7} {'queue':>6}  kernel")
for n, s, e, g, w, q in rows:
    if s < t0:
        continue
    if "attn_fwd" in n:
        print(f"{(s - t0) / 1e3:9.1f}  first decoder-layer attention; kernel time {tot / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us")
        break
    gap = max(0, s - last)
    print(f"{(s - t0) / 1e3:9.1f} {gap / 1e3:8.1f} {(e - s) / 1e3:8.1f} {int(g) // max(int(w), 1):7d} {q!s:>6}  {n.split('(')[0][-80:]}")
    tot += e - s
    gaps += gap
    last = max(last, e)
