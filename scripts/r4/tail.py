#!/usr/bin/env python3
"""The tail of a TF step from a rocprofv3 kernel trace of bench.py: what runs after the main queue's last backward kernel.  python scripts/r4/tail.py <dir>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ce = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('softmax_ce')]
for k in (-4, -3, -2):                                   # three steps: from one loss kernel to the next
    seg = rows[ce[k]:ce[k + 1]]
    ad = [r for r in seg if r['Kernel_Name'].startswith('adamw_kernel')]
    last_adam = ad[-1]
    step = [r for r in seg if r['s'] <= last_adam['s']]
    qs = {}
    for r in step: qs.setdefault(r['Queue_Id'], []).append(r)
    main = max(qs.values(), key=len)
    side = sorted([r for l in qs.values() if l is not main for r in l], key=lambda r: r['s'])
    main_bwd = [r for r in main if not r['Kernel_Name'].startswith('adamw') and 'elementwise' not in r['Kernel_Name'] and 'increment' not in r['Kernel_Name']]
    m_end = main_bwd[-1]['e']
    t_end = last_adam['s']
    tail = [r for r in side if r['e'] > m_end]
    print(f"step: main queue's last backward kernel ends {(t_end - m_end) / 1e3:7.1f} us before the final AdamW starts; side-queue kernels in that window:")
    for r in tail:
        print(f"     {r['Kernel_Name'][:44]:44s} start {(r['s'] - m_end) / 1e3:8.1f} us  dur {(r['e'] - r['s']) / 1e3:7.1f} us")
