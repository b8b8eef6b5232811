#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 results database -> CSV on stdout.   usage: rocprof_summary.py results.db [per]
`per` divides the call counts / totals (e.g. the number of decoded tokens) for a per-unit column."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
agg = collections.defaultdict(lambda: [0, 0])
for st, en, name in c.execute(f"select k.start, k.end, s.kernel_name from {kd} k join {sym} s on k.kernel_id = s.id"):
    a = agg[name]; a[0] += 1; a[1] += en - st
tot = sum(v[1] for v in agg.values())
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","CallsPerUnit","UsPerUnit"')
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{name}",{n},{t},{t / n:.1f},{100.0 * t / tot:.2f},{n / per:.3f},{t / per / 1000:.2f}')
