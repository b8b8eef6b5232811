#!/usr/bin/env python3
"""Kernel sequence (name, start offset us, duration us) of a slice of a rocprofv3 results database: rocprof_sequence.py results.db [first] [count]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 0; count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select k.start, k.end, s.kernel_name from {kd} k join {sym} s on k.kernel_id = s.id order by k.start"))
t0 = rows[first][0]
prev_end = t0
for st, en, name in rows[first:first + count]:
    print(f"{(st - t0) / 1e3:10.2f} us  gap {(st - prev_end) / 1e3:6.2f}  dur {(en - st) / 1e3:7.2f}  {name[:90]}")
    prev_end = en
