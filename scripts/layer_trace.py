#!/usr/bin/env python3
"""Consecutive main-queue kernels of one CvT stage-3 layer in backward (durations and gaps) from a rocprofv3 kernel trace: python scripts/layer_trace.py <dir>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
byq = collections.defaultdict(list)
for r in rows: byq[r['Queue_Id']].append(r)
main = max(byq.values(), key=len)
# last step: find the last adamw, walk back to a dw3_dx (end of a layer's projection backward), print the ~45 kernels before the 3rd-from-last dx
idx = [i for i, r in enumerate(main) if 'dw3_dx' in r['Kernel_Name']]
end = idx[-21 * 2 - 8]
beg = idx[-21 * 2 - 9]
prev = main[beg]['e']
tot = gap = 0
for r in main[beg + 1:end + 1]:
    g = r['s'] - prev
    print(f"{r['Kernel_Name'][:58]:58s} dur {(r['e'] - r['s']) / 1e3:7.1f} us  gap {g / 1e3:6.1f} us")
    tot += r['e'] - r['s']; gap += max(0, g); prev = r['e']
print(f"layer: kernels {tot / 1e3:.1f} us, gaps {gap / 1e3:.1f} us")
