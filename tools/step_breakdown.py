"""Per-kernel time inside ONE training step of a rocprofv3 rocpd trace (model init and warm-up excluded).
A step = from the end of one optimizer burst (adamw_vec4_kernel launches closer than 5 ms to each other) to the end of the next.
Usage: python tools/step_breakdown.py <results.db> [step_index_from_end=1] [top=40]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
opt = [(s, e) for n, s, e in rows if "adamw" in n]
bursts, cb = [], [opt[0]]
for h in opt[1:]:
    if h[0] - cb[-1][1] > 5e6:
        bursts.append(cb)
        cb = [h]
    else:
        cb.append(h)
bursts.append(cb)
assert len(bursts) >= back + 1, f"only {len(bursts)} optimizer bursts in the trace"
t0, t1 = bursts[-back - 1][-1][1], bursts[-back][-1][1]
inside = [(n, s, e) for n, s, e in rows if s >= t0 and e <= t1]
busy, last_end, gaps = 0.0, t0, 0.0
per = defaultdict(lambda: [0, 0.0])
for n, s, e in inside:
    per[n.split("(")[0][-70:]][0] += 1
    per[n.split("(")[0][-70:]][1] += e - s
    if s > last_end:
        gaps += s - last_end
    last_end = max(last_end, e)
busy = sum(v[1] for v in per.values())
gl, le, ln = [], t0, "(step start)"
for n, s_, e_ in inside:
    if s_ > le:
        gl.append((s_ - le, (le - t0) / 1e6, ln, n.split("(")[0][-40:]))
    if e_ > le:
        le, ln = e_, n.split("(")[0][-40:]
hist = defaultdict(lambda: [0, 0.0])
for g_, *_ in gl:
    b_ = "<5us" if g_ < 5e3 else "<20us" if g_ < 20e3 else "<100us" if g_ < 100e3 else ">=100us"
    hist[b_][0] += 1
    hist[b_][1] += g_
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, {len(inside)} launches, kernel time {busy / 1e6:.2f} ms, idle gaps {gaps / 1e6:.2f} ms")
for name, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {t / 1e6:8.2f} ms {c:5d} x {t / c / 1e3:9.1f} us  {name}")
print("gaps by size:", {k: (v[0], round(v[1] / 1e6, 2)) for k, v in hist.items()})
for g_, at, a_, b_ in sorted(gl, key=lambda x: -x[0])[:12]:
    print(f"  gap {g_ / 1e3:8.1f} us at {at:8.2f} ms  after {a_}  before {b_}")
byprev, bynext = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
for g_, at, a_, b_ in gl:
    byprev[a_][0] += 1; byprev[a_][1] += g_
    bynext[b_][0] += 1; bynext[b_][1] += g_
cnt = defaultdict(int)
for n, s_, e_ in inside:
    cnt[n.split("(")[0][-40:]] += 1
print("gaps grouped by the kernel BEFORE the gap (gaps / launches of that kernel, total us):")
for k, v in sorted(byprev.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {v[0]:4d} / {cnt[k]:4d}  {v[1] / 1e3:8.1f} us  {k}")
print("gaps grouped by the kernel AFTER the gap:")
for k, v in sorted(bynext.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {v[0]:4d} / {cnt[k]:4d}  {v[1] / 1e3:8.1f} us  {k}")
import os
if os.environ.get("SEQ"):
    print("first launches of the step:")
    frm = float(os.environ.get("SEQ_FROM_MS", "0")) * 1e6
    for n, s_, e_ in [r for r in inside if r[1] - t0 >= frm][:int(os.environ["SEQ"])]:
        print(f"  {(s_ - t0) / 1e3:9.1f} us +{(e_ - s_) / 1e3:7.1f}  {n.split('(')[0][-80:]}")
