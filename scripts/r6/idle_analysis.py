#!/usr/bin/env python3
"""Round 6: GPU idle time inside an SCST step from a rocprofv3 --kernel-trace CSV: steps are cut at the AdamW launch; idle = wall - union of all kernel
intervals (every queue); the largest idle intervals with the kernels on either side.  python3 scripts/r6/idle_analysis.py <kernel_trace.csv> [marker]"""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw"
lo_ms, hi_ms = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.0, 1e9)
for r in rows:
    r["s"], r["e"], r["n"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]
rows.sort(key=lambda r: r["s"])
cut = [i for i, r in enumerate(rows) if marker in r["n"]]
# one cut per step: keep the last marker of a burst (markers closer than 1 ms belong together)
cuts = [c for c, d in zip(cut, cut[1:] + [None]) if d is None or rows[d]["s"] - rows[c]["s"] > 1e6]
out, big, cnt = [], collections.Counter(), collections.Counter()
for a, b in zip(cuts[2:-1], cuts[3:]):
    seg = rows[a + 1:b + 1]
    wall = seg[-1]["e"] - rows[a]["e"]
    if not (lo_ms <= wall / 1e6 <= hi_ms):
        continue
    end, idle = rows[a]["e"], 0
    prev = rows[a]
    for r in seg:
        if r["s"] > end:
            g = r["s"] - end
            idle += g
            if g > 20000:
                key = (prev["n"][:46], r["n"][:46])
                big[key] += g; cnt[key] += 1
        if r["e"] > end:
            end, prev = r["e"], r
    out.append((wall / 1e6, idle / 1e6, len(seg)))
n = len(out)
print("per step (wall ms, idle ms):", " ".join(f"({o[0]:.1f},{o[1]:.2f})" for o in out))
print(f"steps {n}: wall median {statistics.median(o[0] for o in out):.2f} ms, GPU idle (no kernel on any queue) median {statistics.median(o[1] for o in out):.2f} ms, kernels {statistics.median(o[2] for o in out)}")
print("idle intervals > 20 us, per step, by (last kernel before, first kernel after):")
for key, g in big.most_common(30):
    print(f"  {g / 1e3 / n:9.1f} us/step  x{cnt[key] / n:6.1f}  {key[0]:46s} -> {key[1]}")
