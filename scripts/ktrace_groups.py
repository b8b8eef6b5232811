#!/usr/bin/env python3
"""Group a rocprofv3 kernel trace by (kernel, grid): python scripts/ktrace_groups.py <trace.csv> <steps> [name filter]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]); flt = sys.argv[3] if len(sys.argv) > 3 else ""
d = collections.defaultdict(list)
for r in rows:
    if flt in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:50], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print(f"{k[0]:50s} grid {k[1]:>9s} wg {k[2]:>5s} calls/step {len(v)/n:6.1f}  avg {sum(v)/len(v)/1e3:8.1f} us  ms/step {sum(v)/n/1e6:7.3f}")
print("total ms/step", tot / n / 1e6)
