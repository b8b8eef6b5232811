#!/usr/bin/env python3
"""Round 6: idle time of the MAIN queue inside a TF step, from a rocprofv3 --kernel-trace CSV: per step (softmax_ce to softmax_ce) the gaps between one
kernel's end and the next kernel's start on the queue the loss kernel runs on; the largest gaps with the kernels on either side.
python3 scripts/r6/gap_analysis.py <kernel_trace.csv>"""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"], r["n"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]
rows.sort(key=lambda r: r["s"])
ce = [i for i, r in enumerate(rows) if "softmax_ce" in r["n"]]
mainq = rows[ce[len(ce) // 2]]["Queue_Id"]
tot, big = [], collections.Counter()
examples = {}
hist = collections.Counter()
for a, b in zip(ce[3:-1], ce[4:]):
    main = [r for r in rows[a:b] if r["Queue_Id"] == mainq]
    wall = main[-1]["s"] - main[0]["s"]
    end, gsum = main[0]["e"], 0
    for prev, cur in zip(main, main[1:]):
        end = max(end, prev["e"])
        g = cur["s"] - end
        if g > 0:
            gsum += g
            hist[min(int(g / 1000), 20)] += 1
            if g > 3000:
                key = (prev["n"][:44], cur["n"][:44])
                big[key] += g
                examples[key] = examples.get(key, 0) + 1
    tot.append((gsum / 1e3, wall / 1e3, len(main)))
n = len(tot)
print(f"steps {n}: main-queue idle per step: median {statistics.median(t[0] for t in tot):.1f} us of {statistics.median(t[1] for t in tot):.1f} us wall, {statistics.median(t[2] for t in tot)} kernels")
print("gap histogram (us -> count per step):", {k: round(v / n, 1) for k, v in sorted(hist.items())})
print("gaps > 3 us, summed per step, by (kernel before, kernel after):")
for key, g in big.most_common(25):
    print(f"  {g / 1e3 / n:8.1f} us/step  x{examples[key] / n:5.1f}  {key[0]:44s} -> {key[1]}")
