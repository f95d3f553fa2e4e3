"""Print the dispatch sequence between the n-th and (n+1)-th occurrence of a marker kernel (rocprofv3 rocpd database).
Usage: python tools/trace_seq.py <results.db> <marker substr> <n>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
n = int(sys.argv[3])
a, b = idx[n], idx[n + 1]
for name, s, e in rows[a:b + 1]:
    print(f"{(s - rows[a][1]) / 1e3:10.1f} us +{(e - s) / 1e3:8.1f}  {name.split('(')[0][-70:]}")
