#!/usr/bin/env python3
"""GPU idle time inside the SCST steps of a rocprofv3 kernel trace of bench.py: python scripts/r4/scst_gaps.py <dir>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# an SCST step ends with its AdamW; steps are told apart from TF steps by the decode kernels between two AdamW launches
ad = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adamw_kernel')]
steps = []
for a, b in zip(ad, ad[1:]):
    seg = rows[a + 1:b + 1]
    if sum(1 for r in seg if 'dec_gemm_kernel' in r['Kernel_Name']) > 5000:
        steps.append(seg)
for seg in steps[-6:]:
    t0, t1 = seg[0]['s'], seg[-1]['e']
    ev = sorted([(r['s'], 1) for r in seg] + [(r['e'], -1) for r in seg])
    cur, last, idle, gaps = 0, t0, 0, []
    for t, d in ev:
        if cur == 0 and t > last:
            idle += t - last
            if t - last > 30000: gaps.append((t - last, t))
        cur += d; last = t
    dec = [r for r in seg if 'dec_gemm_kernel' in r['Kernel_Name']]
    d0, d1 = dec[0]['s'], dec[-1]['e']
    print(f"step {(t1 - t0) / 1e6:7.2f} ms: decode window {(d1 - d0) / 1e6:6.2f} ms, before it {(d0 - t0) / 1e6:5.2f} ms, after it {(t1 - d1) / 1e6:5.2f} ms; GPU idle {idle / 1e6:5.2f} ms "
          f"(gaps > 30 us: {len(gaps)}, {sum(g[0] for g in gaps) / 1e6:.2f} ms)")
    for g, t in sorted(gaps, reverse=True)[:6]:
        nxt = next(r for r in seg if r['s'] >= t)
        where = "before decode" if t < d0 else ("inside decode" if t < d1 else "after decode")
        print(f"      {g / 1e3:8.1f} us  {where:14s} before {nxt['Kernel_Name'][:50]}")
