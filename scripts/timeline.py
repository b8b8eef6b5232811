#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of bench.py: python scripts/timeline.py gpurun_out/profX"""
import csv, glob, collections, statistics, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adamw_kernel')]
i0, i1 = ad[-3] + 1, ad[-2] + 1
seg = rows[i0:i1]
t0, t1 = seg[0]['s'], seg[-1]['e']
print('step span ms', (t1 - t0) / 1e6, 'kernels', len(seg))
byq = collections.defaultdict(list)
for r in seg: byq[r['Queue_Id']].append(r)
for q, lst in byq.items():
    print('queue', q, 'kernels', len(lst), 'busy ms', sum(r['e'] - r['s'] for r in lst) / 1e6)
ev = sorted([(r['s'], 1) for r in seg] + [(r['e'], -1) for r in seg])
cur = 0; last = t0; idle = 0; both = 0
for t, d in ev:
    if cur == 0: idle += t - last
    if cur >= 2: both += t - last
    cur += d; last = t
print('idle ms', idle / 1e6, 'overlap ms', both / 1e6)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    agg = collections.Counter(); cnt = collections.Counter()
    for r in lst: agg[r['Kernel_Name'][:48]] += r['e'] - r['s']; cnt[r['Kernel_Name'][:48]] += 1
    print('--- queue', q)
    for k, v in agg.most_common(12): print(f"  {k:50s} {cnt[k]:5d} {v/1e6:8.3f} ms  avg {v/cnt[k]/1e3:7.1f} us")
# main-queue idle gaps (the queue with the most kernels): how long the critical stream waited, and for what
main = max(byq.items(), key=lambda kv: len(kv[1]))[1]
gaps = []
for a, b in zip(main, main[1:]):
    if b['s'] > a['e']:
        gaps.append((b['s'] - a['e'], a['Kernel_Name'][:40], b['Kernel_Name'][:40]))
tot = sum(g[0] for g in gaps)
print(f"main-queue gaps: {len(gaps)} totalling {tot/1e6:.3f} ms; >20us: {sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000)/1e6:.3f} ms")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"   {g[0]/1e3:8.1f} us after {g[1]:40s} before {g[2]}")
hist = collections.Counter(min(int(g[0] / 1000), 20) for g in gaps)
print("gap histogram (us: count):", dict(sorted(hist.items())))
# tail of the weight-gradient queue: side-queue work that runs after the main queue's last backward kernel (the optimiser waits for both)
if len(byq) >= 2:
    qs = sorted(byq.items(), key=lambda kv: -len(kv[1]))
    mainq, sideq = qs[0][1], qs[1][1]
    side_end = max(r['e'] for r in sideq)
    before = [r for r in mainq if r['e'] <= side_end and not r['Kernel_Name'].startswith('adamw')]
    after = [r for r in mainq if r['s'] >= side_end]
    last_main = max((r['e'] for r in before), default=t0)
    print(f"side-queue end - last main-queue kernel end before it: {(side_end - last_main) / 1e3:.1f} us; first main kernel after the side queue ends: "
          f"{after[0]['Kernel_Name'][:40] if after else None} at +{(after[0]['s'] - side_end) / 1e3 if after else 0:.1f} us")
    # side-queue idle time inside the backward window (from its first kernel to its last)
    s0 = min(r['s'] for r in sideq)
    busy = sum(r['e'] - r['s'] for r in sideq)
    print(f"side queue: window {(side_end - s0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
    tailk = sorted(sideq, key=lambda r: r['e'])[-6:]
    for r in tailk: print(f"   side tail: {r['Kernel_Name'][:50]:50s} start +{(r['s'] - last_main) / 1e3:8.1f} us  dur {(r['e'] - r['s']) / 1e3:7.1f} us")
